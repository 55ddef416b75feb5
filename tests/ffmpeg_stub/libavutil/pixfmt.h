#pragma once
enum AVPixelFormat { AV_PIX_FMT_NONE = -1, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV420P10LE, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P,
    AV_PIX_FMT_YUV422P10LE, AV_PIX_FMT_YUV444P10LE, AV_PIX_FMT_GRAY8, AV_PIX_FMT_NV12, AV_PIX_FMT_P010, AV_PIX_FMT_VAAPI, AV_PIX_FMT_DRM_PRIME };
typedef struct AVComponentDescriptor { int plane, step, offset, shift, depth; } AVComponentDescriptor;
typedef struct AVPixFmtDescriptor { int nb_components, log2_chroma_w, log2_chroma_h; AVComponentDescriptor comp[4]; } AVPixFmtDescriptor;
const AVPixFmtDescriptor *av_pix_fmt_desc_get(int fmt);
