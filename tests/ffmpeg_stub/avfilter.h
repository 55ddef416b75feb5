#pragma once
#include <string.h>
#include <errno.h>
#include "libavutil/opt.h"
#include "libavutil/pixfmt.h"
#include "libavutil/hwcontext.h"
#define av_cold
#define AVERROR(e) (-(e))
#define AVERROR_EOF (-0x20464f45)
#define AV_LOG_ERROR 16
#define AV_LOG_INFO 32
#define AV_LOG_VERBOSE 40
#define AV_CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))
#define NULL_IF_CONFIG_SMALL(x) x
#define AVFILTER_FLAG_SUPPORT_TIMELINE_GENERIC (1 << 16)
enum AVMediaType { AVMEDIA_TYPE_VIDEO };
void av_log(void *avcl, int level, const char *fmt, ...);
typedef struct AVFrame { unsigned char *data[8]; int linesize[8]; unsigned char **extended_data; int width, height, format; AVBufferRef *buf[8]; AVBufferRef *hw_frames_ctx; } AVFrame;
AVFrame *av_frame_alloc(void);
void av_frame_free(AVFrame **f);
int av_frame_copy_props(AVFrame *dst, const AVFrame *src);
struct AVFilterContext; struct AVFilterLink;
typedef struct AVFilterLink { struct AVFilterContext *src, *dst; int w, h, format; AVBufferRef *hw_frames_ctx; } AVFilterLink;
typedef struct AVFilterContext { void *priv; AVFilterLink **inputs, **outputs; } AVFilterContext;
typedef struct AVFilterPad { const char *name; enum AVMediaType type; int (*config_props)(AVFilterLink *);
    int (*filter_frame)(AVFilterLink *, AVFrame *); int (*request_frame)(AVFilterLink *);
    union { AVFrame *(*video)(AVFilterLink *link, int w, int h); } get_buffer; } AVFilterPad;
typedef struct AVFilterFormats AVFilterFormats;
typedef struct AVFilter { const char *name, *description; int priv_size; int (*init)(AVFilterContext *); void (*uninit)(AVFilterContext *);
    const enum AVPixelFormat *pix_fmts; const AVFilterPad *inputs, *outputs; const AVClass *priv_class; int flags, flags_internal; enum AVPixelFormat single_pixfmt; } AVFilter;
#define AVFILTER_DEFINE_CLASS(n) static const AVClass n##_class = { #n, n##_options }
#define FILTER_PIXFMTS_ARRAY(a) .pix_fmts = a
#define FILTER_SINGLE_PIXFMT(f) .single_pixfmt = f
#define FF_FILTER_FLAG_HWFRAME_AWARE 1
#define ENAVAIL_ 119
#define FILTER_INPUTS(a) .inputs = a
#define FILTER_OUTPUTS(a) .outputs = a
