#pragma once
enum AVPixelFormat { AV_PIX_FMT_NONE = -1, AV_PIX_FMT_YUV420P, AV_PIX_FMT_YUV420P10LE, AV_PIX_FMT_YUV422P, AV_PIX_FMT_YUV444P,
    AV_PIX_FMT_YUV422P10LE, AV_PIX_FMT_YUV444P10LE, AV_PIX_FMT_GRAY8 };
typedef struct AVPixFmtDescriptor { int log2_chroma_w, log2_chroma_h; } AVPixFmtDescriptor;
const AVPixFmtDescriptor *av_pix_fmt_desc_get(int fmt);
