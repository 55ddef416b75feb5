/*
 * vf_raisr_hipframes.c -- "raisr_hip": Enhanced RAISR on frames that never leave the GPU.
 *
 * Counterpart of the reference's hardware-frames filter (ffmpeg/vf_raisr_opencl.c: AV_PIX_FMT_OPENCL frames in and out,
 * RNLHandler_SetOpenCLContext + ASMType OpenCLExternal, SetRes on the first frame, Process per frame).  FFmpeg has no HIP
 * hwcontext; on AMD GPUs decoded frames live in VAAPI surfaces, which export as DRM PRIME dma-bufs.  This filter therefore takes
 * AV_PIX_FMT_VAAPI frames, maps input and output surfaces to DRM PRIME (av_hwframe_map), imports the dma-bufs into the HIP
 * address space (hipImportExternalMemory, cached per surface: decoders and filters recycle a small pool) and hands DEVICE pointers
 * to the library: ASMType HIPExternal (include/raisr/RaisrDefaults.h), RNLHandler_SetOpenCLContext(stream, NULL, 0, device).
 *
 *   ffmpeg -hwaccel vaapi -hwaccel_output_format vaapi -i in.mp4 \
 *          -vf "scale_vaapi=format=yuv420p:mode=fast,raisr_hip=ratio=2:filterfolder=filters_2x/filters_highres:passes=2" \
 *          -c:v hevc_vaapi out.mp4
 *
 * Requirements the filter checks instead of assuming: planar surfaces (yuv420p / yuv420p10; NV12 / P010 are accepted as well,
 * their interleaved chroma plane goes through the library's two-channel cheap upscale), LINEAR layout (DRM_FORMAT_MOD_LINEAR --
 * a tiled surface is refused with an explanation: the kernels address rows and pitches), one dma-buf object per layer plane.
 *
 * Build: libavfilter/Makefile  OBJS-$(CONFIG_RAISR_HIP_FILTER) += vf_raisr_hipframes.o ; allfilters.c  extern const AVFilter
 * ff_vf_raisr_hip ; configure  raisr_hip_filter_deps="libraisr vaapi libdrm" , link -lraisr -lamdhip64 -lstdc++.
 * In this repository the file is lint-compiled against declaration stubs (tests/test_ffmpeg_patch.py); the library entry points
 * it calls are exercised on the GPU by tests/test_gpu_host_api.py::test_hipexternal_device_planes_against_the_oracle.
 */
#include <unistd.h>
#include <sys/stat.h>
#include <hip/hip_runtime_api.h>

#include "raisr/RaisrHandler.h"
#include "raisr/RaisrDefaults.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_drm.h"
#include "avfilter.h"
#include "internal.h"
#include "video.h"

#define RAISR_HIP_MAX_IMPORTS 64           /* surfaces remembered (decoder pool + filter pool) */
#define DRM_FORMAT_MOD_LINEAR_ 0ULL

typedef struct ImportedObject {
    /* Identity of the dma-buf.  NOT the descriptor number: av_hwframe_map() exports fresh descriptors per call and
     * av_frame_free() of the mapping closes them, so the kernel hands the same small numbers to different surfaces from the
     * second frame on.  Every dma-buf has its own inode on the dmabuf pseudo filesystem for as long as the buffer lives. */
    dev_t st_dev;
    ino_t st_ino;
    size_t size;
    hipExternalMemory_t mem;
    void *base;                            /* device pointer of the whole object */
} ImportedObject;

typedef struct RaisrHipContext {
    const AVClass *class;
    float ratio;
    int bits;
    int range;
    char *filterfolder;
    int blending;
    int passes;
    int mode;
    int device;
    int evenoutput;

    int initialised;
    enum AVPixelFormat sw_format;
    AVBufferRef *out_frames_ref;           /* VAAPI frames context of the output link */
    hipStream_t stream;
    ImportedObject imports[RAISR_HIP_MAX_IMPORTS];
    int nb_imports;
} RaisrHipContext;

/* device pointer of a dma-buf object, imported once per surface */
static int import_object(AVFilterContext *avctx, int fd, size_t size, void **base)
{
    RaisrHipContext *ctx = avctx->priv;
    hipExternalMemoryHandleDesc hd = { 0 };
    hipExternalMemoryBufferDesc bd = { 0 };
    ImportedObject *io;
    struct stat st;

    if (fstat(fd, &st) < 0) {
        av_log(avctx, AV_LOG_ERROR, "fstat of dma-buf %d failed\n", fd);
        return AVERROR(EINVAL);
    }
    for (int i = 0; i < ctx->nb_imports; i++)
        if (ctx->imports[i].st_ino == st.st_ino && ctx->imports[i].st_dev == st.st_dev && ctx->imports[i].size == size) {
            *base = ctx->imports[i].base;
            return 0;
        }
    if (ctx->nb_imports == RAISR_HIP_MAX_IMPORTS) {          /* forget the oldest surface */
        hipDestroyExternalMemory(ctx->imports[0].mem);
        memmove(&ctx->imports[0], &ctx->imports[1], sizeof(ctx->imports[0]) * (RAISR_HIP_MAX_IMPORTS - 1));
        ctx->nb_imports--;
    }
    io = &ctx->imports[ctx->nb_imports];
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = dup(fd);                                  /* the import takes ownership of the descriptor it is given */
    hd.size = size;
    if (hd.handle.fd < 0 || hipImportExternalMemory(&io->mem, &hd) != hipSuccess) {
        av_log(avctx, AV_LOG_ERROR, "hipImportExternalMemory failed for dma-buf %d (%zu bytes)\n", fd, size);
        return AVERROR(ENOMEM);
    }
    bd.offset = 0;
    bd.size = size;
    if (hipExternalMemoryGetMappedBuffer(&io->base, io->mem, &bd) != hipSuccess) {
        hipDestroyExternalMemory(io->mem);
        av_log(avctx, AV_LOG_ERROR, "hipExternalMemoryGetMappedBuffer failed\n");
        return AVERROR(ENOMEM);
    }
    io->st_dev = st.st_dev;
    io->st_ino = st.st_ino;
    io->size = size;
    ctx->nb_imports++;
    *base = io->base;
    return 0;
}

/* VideoDataType descriptors (device pointers) of a VAAPI frame: map to DRM PRIME, check the layout, import */
static int describe_surface(AVFilterContext *avctx, AVFrame *hw, int writable, AVFrame **mapped, VideoDataType vdt[3], int *interleaved_chroma)
{
    RaisrHipContext *ctx = avctx->priv;
    const AVPixFmtDescriptor *desc = av_pix_fmt_desc_get(ctx->sw_format);
    const AVDRMFrameDescriptor *drm;
    AVFrame *m = av_frame_alloc();
    int err, np = 0;

    if (!m || !desc)
        return AVERROR(ENOMEM);
    m->format = AV_PIX_FMT_DRM_PRIME;
    err = av_hwframe_map(m, hw, writable ? AV_HWFRAME_MAP_WRITE | AV_HWFRAME_MAP_OVERWRITE : AV_HWFRAME_MAP_READ);
    if (err < 0) {
        av_log(avctx, AV_LOG_ERROR, "cannot export the surface as DRM PRIME (%d)\n", err);
        av_frame_free(&m);
        return err;
    }
    drm = (const AVDRMFrameDescriptor *)m->data[0];
    for (int l = 0; l < drm->nb_layers; l++)
        for (int p = 0; p < drm->layers[l].nb_planes && np < 3; p++, np++) {
            const AVDRMPlaneDescriptor *pl = &drm->layers[l].planes[p];
            const AVDRMObjectDescriptor *ob = &drm->objects[pl->object_index];
            void *base;
            if (ob->format_modifier != DRM_FORMAT_MOD_LINEAR_) {
                av_log(avctx, AV_LOG_ERROR, "surface is tiled (modifier %#llx): the RAISR kernels address rows and pitches -- "
                       "allocate linear surfaces (e.g. scale_vaapi / hwupload with a linear frames context)\n",
                       (unsigned long long)ob->format_modifier);
                av_frame_free(&m);
                return AVERROR(ENOSYS);
            }
            err = import_object(avctx, ob->fd, ob->size, &base);
            if (err < 0) {
                av_frame_free(&m);
                return err;
            }
            vdt[np].pData = (unsigned char *)base + pl->offset;
            vdt[np].step = (unsigned int)pl->pitch;
            vdt[np].width = np ? AV_CEIL_RSHIFT(hw->width, desc->log2_chroma_w) : hw->width;
            vdt[np].height = np ? AV_CEIL_RSHIFT(hw->height, desc->log2_chroma_h) : hw->height;
            vdt[np].bitShift = desc->comp[np < desc->nb_components ? np : 0].shift;
        }
    /* NV12 / P010: one chroma plane of interleaved (U, V) pairs -> two descriptors on the same plane, one sample apart;
     * bitShift's top bit tells the library that consecutive samples of a plane are two samples apart (RaisrDefaults.h) */
    *interleaved_chroma = np == 2;
    if (np == 2) {
        const unsigned bps = ctx->bits > 8 ? 2 : 1;
        vdt[2] = vdt[1];
        vdt[2].pData = vdt[1].pData + bps;
        vdt[1].bitShift |= RAISR_HIP_INTERLEAVED2;
        vdt[2].bitShift |= RAISR_HIP_INTERLEAVED2;
    } else if (np != 3) {
        av_log(avctx, AV_LOG_ERROR, "unsupported surface layout (%d planes)\n", np);
        av_frame_free(&m);
        return AVERROR(EINVAL);
    }
    *mapped = m;
    return 0;
}

static int raisr_hip_filter_frame(AVFilterLink *inlink, AVFrame *input)
{
    AVFilterContext *avctx = inlink->dst;
    AVFilterLink *outlink = avctx->outputs[0];
    RaisrHipContext *ctx = avctx->priv;
    AVFrame *output = NULL, *map_in = NULL, *map_out = NULL;
    VideoDataType vdt_in[3] = { 0 }, vdt_out[3] = { 0 };
    int err, il_in = 0, il_out = 0;
    RNLERRORTYPE ret;

    if (!input->hw_frames_ctx) {
        err = AVERROR(EINVAL);
        goto fail;
    }
    output = av_frame_alloc();
    if (!output) {
        err = AVERROR(ENOMEM);
        goto fail;
    }
    err = av_hwframe_get_buffer(ctx->out_frames_ref, output, 0);
    if (err < 0)
        goto fail;
    err = describe_surface(avctx, input, 0, &map_in, vdt_in, &il_in);
    if (err < 0)
        goto fail;
    err = describe_surface(avctx, output, 1, &map_out, vdt_out, &il_out);
    if (err < 0)
        goto fail;
    if (il_in != il_out) {
        err = AVERROR(EINVAL);
        goto fail;
    }
    if (!ctx->initialised) {
        ret = RNLHandler_SetRes(&vdt_in[0], &vdt_in[1], &vdt_in[2], &vdt_out[0], &vdt_out[1], &vdt_out[2]);
        if (ret != RNLErrorNone) {
            av_log(avctx, AV_LOG_ERROR, "RNLHandler_SetRes error\n");
            err = AVERROR(ENOMEM);
            goto fail;
        }
        ctx->initialised = 1;
    }
    /* stream-ordered: Process enqueues on ctx->stream and returns; the surfaces are handed on only after the stream has drained
     * (VAAPI consumers synchronise on the surface, not on a HIP stream) */
    ret = RNLHandler_Process(&vdt_in[0], &vdt_in[1], &vdt_in[2], &vdt_out[0], &vdt_out[1], &vdt_out[2], ctx->blending);
    if (ret != RNLErrorNone || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        av_log(avctx, AV_LOG_ERROR, "RNLHandler_Process error\n");
        err = AVERROR(ENOMEM);
        goto fail;
    }
    err = av_frame_copy_props(output, input);
    if (err < 0)
        goto fail;
    av_frame_free(&map_in);
    av_frame_free(&map_out);
    av_frame_free(&input);
    return ff_filter_frame(outlink, output);

fail:
    av_frame_free(&map_in);
    av_frame_free(&map_out);
    av_frame_free(&input);
    av_frame_free(&output);
    return err;
}

static int raisr_hip_config_input(AVFilterLink *inlink)
{
    AVHWFramesContext *input_frames;

    if (!inlink->hw_frames_ctx)
        return AVERROR(EINVAL);
    input_frames = (AVHWFramesContext *)inlink->hw_frames_ctx->data;
    if (input_frames->format != AV_PIX_FMT_VAAPI)
        return AVERROR(EINVAL);
    if (input_frames->sw_format != AV_PIX_FMT_YUV420P && input_frames->sw_format != AV_PIX_FMT_YUV420P10LE &&
        input_frames->sw_format != AV_PIX_FMT_NV12 && input_frames->sw_format != AV_PIX_FMT_P010)
        return AVERROR(EINVAL);
    return 0;
}

static int raisr_hip_config_output(AVFilterLink *outlink)
{
    AVFilterContext *avctx = outlink->src;
    AVFilterLink *inlink = avctx->inputs[0];
    RaisrHipContext *ctx = avctx->priv;
    AVHWFramesContext *input_frames = (AVHWFramesContext *)inlink->hw_frames_ctx->data, *out_frames;
    const AVPixFmtDescriptor *desc;
    RNLERRORTYPE ret;
    int err;

    ctx->sw_format = input_frames->sw_format;
    desc = av_pix_fmt_desc_get(ctx->sw_format);
    if (desc && desc->comp[0].depth != ctx->bits) {
        av_log(avctx, AV_LOG_ERROR, "input pixel doesn't match model's bitdepth\n");
        return AVERROR(EINVAL);
    }
    outlink->w = inlink->w * ctx->ratio;
    outlink->h = inlink->h * ctx->ratio;
    if (ctx->evenoutput == 1) {
        outlink->w -= outlink->w % 2;
        outlink->h -= outlink->h % 2;
    }
    /* output surfaces: same device, same software format, the upscaled size */
    av_buffer_unref(&ctx->out_frames_ref);
    ctx->out_frames_ref = av_hwframe_ctx_alloc(input_frames->device_ref);
    if (!ctx->out_frames_ref)
        return AVERROR(ENOMEM);
    out_frames = (AVHWFramesContext *)ctx->out_frames_ref->data;
    out_frames->format = AV_PIX_FMT_VAAPI;
    out_frames->sw_format = ctx->sw_format;
    out_frames->width = outlink->w;
    out_frames->height = outlink->h;
    out_frames->initial_pool_size = 8;
    err = av_hwframe_ctx_init(ctx->out_frames_ref);
    if (err < 0)
        return err;
    av_buffer_unref(&outlink->hw_frames_ctx);
    outlink->hw_frames_ctx = av_buffer_ref(ctx->out_frames_ref);
    if (!outlink->hw_frames_ctx)
        return AVERROR(ENOMEM);

    /* the library on the GPU that owns the surfaces, frames stream-ordered on our stream (what SetOpenCLContext carries for
     * ASMType HIPExternal: the OpenCL filter passes its cl_context / cl_device_id through the same call) */
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
        return AVERROR(ENODEV);
    ret = RNLHandler_SetOpenCLContext(ctx->stream, NULL, 0, ctx->device);
    if (ret == RNLErrorNone)
        ret = RNLHandler_Init(ctx->filterfolder, ctx->ratio, ctx->bits, (RangeType)ctx->range, 1, HIPExternal, ctx->passes, ctx->mode);
    if (ret != RNLErrorNone) {
        av_log(avctx, AV_LOG_ERROR, "RNLHandler_Init failed\n");
        return AVERROR(ENAVAIL);
    }
    ctx->initialised = 0;
    return 0;
}

static av_cold void raisr_hip_uninit(AVFilterContext *avctx)
{
    RaisrHipContext *ctx = avctx->priv;

    RNLHandler_Deinit();
    for (int i = 0; i < ctx->nb_imports; i++)
        hipDestroyExternalMemory(ctx->imports[i].mem);
    ctx->nb_imports = 0;
    if (ctx->stream)
        hipStreamDestroy(ctx->stream);
    ctx->stream = NULL;
    av_buffer_unref(&ctx->out_frames_ref);
}

#define OFFSET(x) offsetof(RaisrHipContext, x)
#define FLAGS (AV_OPT_FLAG_FILTERING_PARAM | AV_OPT_FLAG_VIDEO_PARAM)
static const AVOption raisr_hip_options[] = {
    {"ratio", "ratio of the upscaling, between 1 and 2", OFFSET(ratio), AV_OPT_TYPE_FLOAT, {.dbl = 2}, 1, 2, FLAGS},
    {"bits", "bit depth", OFFSET(bits), AV_OPT_TYPE_INT, {.i64 = 8}, 8, 10, FLAGS},
    {"range", "input color range (1: video, 2: full)", OFFSET(range), AV_OPT_TYPE_INT, {.i64 = VideoRange}, VideoRange, FullRange, FLAGS},
    {"filterfolder", "absolute filter folder path", OFFSET(filterfolder), AV_OPT_TYPE_STRING, {.str = "filters_2x/filters_lowres"}, 0, 0, FLAGS},
    {"blending", "CT blending mode (1: Randomness, 2: CountOfBitsChanged)", OFFSET(blending), AV_OPT_TYPE_INT, {.i64 = CountOfBitsChanged}, Randomness, CountOfBitsChanged, FLAGS},
    {"passes", "passes to run (1: one pass, 2: two pass)", OFFSET(passes), AV_OPT_TYPE_INT, {.i64 = 1}, 1, 2, FLAGS},
    {"mode", "mode for two pass (1: upscale in 1st pass, 2: upscale in 2nd pass)", OFFSET(mode), AV_OPT_TYPE_INT, {.i64 = 1}, 1, 2, FLAGS},
    {"device", "HIP device ordinal of the GPU that owns the surfaces", OFFSET(device), AV_OPT_TYPE_INT, {.i64 = 0}, 0, INT_MAX, FLAGS},
    {"evenoutput", "make output size as even number (0: ignore, 1: subtract 1px if needed)", OFFSET(evenoutput), AV_OPT_TYPE_INT, {.i64 = 0}, 0, 1, FLAGS},
    {NULL}
};

AVFILTER_DEFINE_CLASS(raisr_hip);

static const AVFilterPad raisr_hip_inputs[] = {
    {
        .name         = "default",
        .type         = AVMEDIA_TYPE_VIDEO,
        .filter_frame = &raisr_hip_filter_frame,
        .config_props = &raisr_hip_config_input,
    }
};

static const AVFilterPad raisr_hip_outputs[] = {
    {
        .name         = "default",
        .type         = AVMEDIA_TYPE_VIDEO,
        .config_props = &raisr_hip_config_output,
    }
};

const AVFilter ff_vf_raisr_hip = {
    .name           = "raisr_hip",
    .description    = NULL_IF_CONFIG_SMALL("Enhanced RAISR on GPU-resident (VAAPI) frames through HIP."),
    .priv_size      = sizeof(RaisrHipContext),
    .priv_class     = &raisr_hip_class,
    .uninit         = &raisr_hip_uninit,
    FILTER_INPUTS(raisr_hip_inputs),
    FILTER_OUTPUTS(raisr_hip_outputs),
    FILTER_SINGLE_PIXFMT(AV_PIX_FMT_VAAPI),
    .flags_internal = FF_FILTER_FLAG_HWFRAME_AWARE,
};
