#pragma once
/* declaration-only subset for the lint compile of ffmpeg/vf_raisr_hipframes.c */
#include "pixfmt.h"
typedef struct AVBufferRef { unsigned char *data; } AVBufferRef;
typedef struct AVHWFramesContext { AVBufferRef *device_ref; enum AVPixelFormat format, sw_format; int width, height, initial_pool_size; } AVHWFramesContext;
enum { AV_HWFRAME_MAP_READ = 1, AV_HWFRAME_MAP_WRITE = 2, AV_HWFRAME_MAP_OVERWRITE = 4 };
struct AVFrame;
int av_hwframe_map(struct AVFrame *dst, const struct AVFrame *src, int flags);
int av_hwframe_get_buffer(AVBufferRef *hwframe_ctx, struct AVFrame *frame, int flags);
AVBufferRef *av_hwframe_ctx_alloc(AVBufferRef *device_ctx);
int av_hwframe_ctx_init(AVBufferRef *ref);
AVBufferRef *av_buffer_ref(const AVBufferRef *buf);
void av_buffer_unref(AVBufferRef **buf);
