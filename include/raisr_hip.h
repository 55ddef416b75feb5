/*
 * raisr_hip.h -- thin C ABI between host code and the MI355X (gfx950) HIP kernels of the
 * Enhanced-RAISR Y-plane hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * This is the device boundary that replaces the reference's per-band CPU hot loop:
 *   raisr_hip_process_*      <->  processSegment()            reference Library/Raisr.cpp:890-1289
 *                                 as fanned out by RNLProcess  Library/Raisr.cpp:1294-1397
 *   raisr_hip_resize_plane   <->  IPPResize(8|16) call sites   Library/Raisr.cpp:947-958,1373-1388
 *   raisr_hip_set_model      <->  result of ReadTrainedData    Library/Raisr.cpp:246-433
 *   raisr_hip_configure      <->  RNLInit parameter state      Library/Raisr.cpp:1409-1679
 *                                 + RNLSetRes resources         Library/Raisr.cpp:1681-1829
 *   raisr_hip_plan_bands     <->  band/zone arithmetic         Library/Raisr.cpp:1738-1779
 * The reference-compatible C API (RNLHandler_*, include/raisr/RaisrHandler.h) is implemented on top of
 * these entry points in csrc/raisr_api.cpp; a foreign-language binding (ctypes, cgo, JNI) can bind
 * either layer.
 *
 * All functions return 0 on success, a negative RAISR_HIP_E* code otherwise, and never throw.
 * A context owns one HIP stream-ordered set of scratch planes ("lane"): use one context per
 * in-flight frame to overlap frames; contexts on one device may be used from different threads.
 */
#ifndef RAISR_HIP_H
#define RAISR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAISR_HIP_OK            0
#define RAISR_HIP_EINVAL      (-1)   /* bad argument / unsupported configuration */
#define RAISR_HIP_ENODEV      (-2)   /* no usable gfx950 device / HIP runtime failure at init */
#define RAISR_HIP_ENOMEM      (-3)   /* device allocation failed */
#define RAISR_HIP_ESTATE      (-4)   /* call order violated (e.g. process before configure) */
#define RAISR_HIP_ERUNTIME    (-5)   /* a HIP call failed; see raisr_hip_last_error() */

/* Hash-variant selection, mirrors the reference's ASMType semantics (RaisrDefaults.h:37-44):
 * which x86 hash flavour the bit-exact path reproduces. */
#define RAISR_HIP_HASH_AVX2     1    /* all columns use the AVX2 hash (rcpps/rsqrtps)           */
#define RAISR_HIP_HASH_AVX512   2    /* AVX-512 hash + AVX2 re-hash of the tail columns          */
#define RAISR_HIP_HASH_FP16     5    /* AVX512-FP16 pipeline (binary16 arithmetic)               */

#define RAISR_HIP_BLEND_RANDOMNESS 1 /* BlendingMode, RaisrDefaults.h:31-35 */
#define RAISR_HIP_BLEND_COUNT      2

#define RAISR_HIP_TIE_HALF_UP   0    /* cheap-upscale rounding of exact .5 ties (build-defined) */
#define RAISR_HIP_TIE_HALF_EVEN 1

typedef struct raisr_hip_ctx raisr_hip_ctx;

typedef struct raisr_hip_config {
    int in_width, in_height;       /* Y plane, input  */
    int out_width, out_height;     /* Y plane, output */
    int bits;                      /* 8, 10 or 16; samples are u8 (bits==8) else u16 LE */
    int clamp_lo, clamp_hi;        /* gMin/gMax of RNLInit (16..235, 64..940, full range) */
    int passes;                    /* 1 or 2 */
    int two_pass_mode;             /* 1: upscale in pass 1; 2: upscale in pass 2 */
    int hash_variant;              /* RAISR_HIP_HASH_* */
    int blending;                  /* RAISR_HIP_BLEND_* */
    int use_pixel_type;            /* 1 when ratio == 2.0 (4 filter phases), else 0 */
    int tie_rule;                  /* RAISR_HIP_TIE_* */
} raisr_hip_config;

/* Lifetime ------------------------------------------------------------------------------------ */
int  raisr_hip_device_count(void);
int  raisr_hip_create(raisr_hip_ctx **out, int device_index);
void raisr_hip_destroy(raisr_hip_ctx *ctx);
const char *raisr_hip_last_error(void);        /* thread-local text of the last failure */
const char *raisr_hip_version(void);

/* Model -------------------------------------------------------------------------------------
 * bank: [hashkeys][pixel_types][121] fp32 in file order (filterbin payload); thresholds are the
 * std::stod() values of the Qfactor tokens (double: the library derives both the (float) and the
 * (_Float16) flavours the reference's fp32 / fp16 loaders use).  pass_index 0 = first pass, 1 = second pass ("_2" files).
 * The packed device blob (fp32 bank padded to 128 taps + binary16 bank + thresholds) can be exported/imported so that
 * one rank loads the files and the others receive the blob by an RCCL broadcast. */
int    raisr_hip_set_model(raisr_hip_ctx *ctx, int pass_index, const float *bank,
                           int hashkeys, int pixel_types, const double qstr[2], const double qcoh[2],
                           int quant_angle);
size_t raisr_hip_model_blob_bytes(int hashkeys, int pixel_types);
int    raisr_hip_pack_model_blob(void *host_blob, const float *bank, int hashkeys, int pixel_types,
                                 const double qstr[2], const double qcoh[2], int quant_angle);
int    raisr_hip_set_model_blob_device(raisr_hip_ctx *ctx, int pass_index, const void *device_blob,
                                       size_t bytes, void *stream);

/* Multi-GPU start-up (one process per GPU): in-place RCCL broadcast of a packed model blob that lives in device
 * memory, from rank `root` to every rank of the communicator `nccl_comm` (an ncclComm_t), stream-ordered on `stream`
 * (hipStream_t or NULL).  Every rank then calls raisr_hip_set_model_blob_device() on its copy.  This is the path's only
 * collective; librccl.so is loaded on first use, so single-GPU consumers carry no RCCL dependency.  The reference is a CPU
 * library and has no counterpart; bench.py / sharding.py issue the same broadcast through torch.distributed. */
int    raisr_hip_broadcast_model_blob(void *nccl_comm, int root, void *device_blob, size_t bytes, void *stream);
/* One process, several GPUs (raisr_hip_stream_create_multi): blobs[0] (on devices[0]) holds the packed model, blobs[i] (on
 * devices[i]) receives it.  Default: n - 1 concurrent peer copies out of devices[0] (xGMI is a full mesh of point-to-point links: the
 * copies to different GPUs run on different links at once; peer access is enabled where the runtime grants it).  RAISR_HIP_RCCL=1 opts
 * into the in-process RCCL path instead (one communicator set per device list, created once per process, grouped ncclBroadcast) --
 * EXPERIMENTAL: it has never run with more than one rank.  RAISR_HIP_NO_RCCL=1 forces the copies.  Synchronous. */
int    raisr_hip_broadcast_model_blob_devices(const int *devices, int n, void *const *device_blobs, size_t bytes);

/* Geometry / resources ------------------------------------------------------------------------ */
int raisr_hip_configure(raisr_hip_ctx *ctx, const raisr_hip_config *cfg);

/* BlendingMode is a per-frame argument of RNLProcess (Raisr.h:28-30): switch it without reallocating.
 * Randomness never writes the pixels [c_final, W-6) of row H-7 (reference behaviour): the device output
 * plane keeps its previous contents there. */
int raisr_hip_set_blending(raisr_hip_ctx *ctx, int blending);

/* MSB-aligned samples of DEVICE frames (P010: 10-bit values stored as value << 6): the device-plane entry points below read
 * sample = stored >> shift and write stored = sample << shift for Y and chroma, as the reference's OpenCL pre/post-process kernels
 * do with VideoDataType::bitShift (Raisr_OpenCL_kernel.h:241-276).  0 (default) = LSB-aligned.  Host-plane entries ignore it,
 * as the reference's CPU path ignores bitShift. */
int raisr_hip_set_sample_shift(raisr_hip_ctx *ctx, int shift);

/* Hot path ------------------------------------------------------------------------------------
 * Device-resident planes.  Pitches are in BYTES.  `stream` is a hipStream_t (NULL = the
 * context's own stream).  Asynchronous: returns after enqueueing. */
int raisr_hip_process_y_device(raisr_hip_ctx *ctx, const void *d_in, size_t in_pitch,
                               void *d_out, size_t out_pitch, void *stream);
/* Frame batch: n Y planes of the configured geometry, device-resident, through ONE launch per kernel (the frame index is the
 * launch's third grid dimension; scratch planes are kept n deep).  For small frames -- 540p, 720p -- whose single launches cannot
 * fill the chip.  The one-launch path needs the planes equally spaced in memory (d_in[i] - d_in[i-1] and d_out[i] - d_out[i-1]
 * constant and positive: the planes of one allocation) and n <= RAISR_HIP_MAX_BATCH; other batches, and pipelines selected by
 * the A/B switches, run frame by frame.  Same bits either way.  The reference has no counterpart (RNLProcess takes one frame,
 * Library/Raisr.cpp:1294); north_star asks for frame batches. */
#define RAISR_HIP_MAX_BATCH 16
int raisr_hip_process_y_device_batch(raisr_hip_ctx *ctx, int n, const void *const *d_in, size_t in_pitch,
                                     void *const *d_out, size_t out_pitch, void *stream);
/* cheap upscale of one plane (chroma path of RNLProcess): src/dst sample type from `bits` */
int raisr_hip_resize_plane_device(raisr_hip_ctx *ctx, const void *d_src, int sw, int sh, size_t spitch,
                                  void *d_dst, int dw, int dh, size_t dpitch, int bits, void *stream);
/* Whole device-resident frame: RAISR on Y plus the cheap upscale of both chroma planes (RNLProcess's work,
 * Raisr.cpp:1369-1389) without leaving HBM -- the zero-copy analogue of ffmpeg/vf_raisr_opencl.c. */
/* ... with element steps (2 = one channel of an interleaved two-channel plane: NV12 / P010 chroma) */
int raisr_hip_resize_plane_device_ex(raisr_hip_ctx *ctx, const void *d_src, int sw, int sh, size_t spitch, int sstep,
                                     void *d_dst, int dw, int dh, size_t dpitch, int dstep, int bits, void *stream);
int raisr_hip_process_frame_device(raisr_hip_ctx *ctx,
                                   const void *d_in_y, size_t in_y_pitch, void *d_out_y, size_t out_y_pitch,
                                   const void *d_in_u, const void *d_in_v, size_t in_c_pitch,
                                   void *d_out_u, void *d_out_v, size_t out_c_pitch,
                                   int chroma_in_w, int chroma_in_h, int chroma_out_w, int chroma_out_h, void *stream);
/* chroma_step = 2: U and V are the two channels of ONE interleaved plane on both sides (d_*_v = d_*_u + one sample) */
int raisr_hip_process_frame_device_ex(raisr_hip_ctx *ctx,
                                      const void *d_in_y, size_t in_y_pitch, void *d_out_y, size_t out_y_pitch,
                                      const void *d_in_u, const void *d_in_v, size_t in_c_pitch,
                                      void *d_out_u, void *d_out_v, size_t out_c_pitch,
                                      int chroma_in_w, int chroma_in_h, int chroma_out_w, int chroma_out_h, int chroma_step, void *stream);
/* Host planes in, host planes out (what RNLProcess hands over): stages through pinned memory,
 * runs Y + both chroma planes, synchronous.  Chroma pointers may be NULL to skip chroma. */
int raisr_hip_process_host(raisr_hip_ctx *ctx,
                           const void *in_y, size_t in_y_pitch, void *out_y, size_t out_y_pitch,
                           const void *in_u, size_t in_u_pitch, void *out_u, size_t out_u_pitch,
                           const void *in_v, size_t in_v_pitch, void *out_v, size_t out_v_pitch,
                           int chroma_in_w, int chroma_in_h, int chroma_out_w, int chroma_out_h);
/* Layout of a PACKED host frame: Y, U, V with tight pitches at offsets[0..2] (256-byte aligned plane starts), total size
 * *total_bytes.  When the three output planes handed to raisr_hip_process_host* / raisr_hip_stream_submit lie like this in
 * one allocation, the frame is downloaded as a single PCIe copy instead of three. */
int raisr_hip_packed_frame_layout(int y_w, int y_h, int c_w, int c_h, int bits, size_t offsets[3], size_t *total_bytes);
int raisr_hip_synchronize(raisr_hip_ctx *ctx);   /* waits for everything the context has enqueued (Y and chroma lanes) */

/* Horizontal bands ----------------------------------------------------------------------------
 * A frame can be cut into horizontal bands that are processed as INDEPENDENT sub-frames, each by its own
 * context (own stream, or own GPU): a band's sub-frame carries enough extra input rows above and below its
 * kept rows that every kept output row equals the whole-frame result bit for bit -- no halo exchange, no
 * ordering between bands.  This is the GPU counterpart of the reference's thread bands (RNLSetRes zone
 * arithmetic, Library/Raisr.cpp:1738-1779; the reference pads bands by gResizeExpand/gHashingExpand rows for
 * the same reason), used (a) by RNLProcess to overlap one band's PCIe transfers with another band's kernels
 * and (b) to split one frame over several GPUs (latency mode).
 *   passes = 0 plans a plane that only goes through the cheap upscale (chroma).
 * Returns the number of bands planned (1..nbands; fewer when the plane is too small or the ratio cannot be
 * aligned), or a negative RAISR_HIP_E* code. */
typedef struct raisr_hip_band {
    int in_row_begin, in_row_count;      /* input rows of the sub-frame (kept rows + padding), frame coordinates   */
    int out_row_begin, out_row_count;    /* output rows the sub-frame produces, frame coordinates                  */
    int keep_begin, keep_count;          /* output rows of the sub-frame that are valid and owned by this band    */
} raisr_hip_band;
int raisr_hip_plan_bands(int in_height, int out_height, int passes, int nbands, raisr_hip_band *bands);

/* raisr_hip_process_host without the final wait and with a row window on the download: only rows
 * [y_skip, y_skip + y_keep) of the Y output (and [c_skip, c_skip + c_keep) of each chroma output) are copied
 * back, to out_y / out_u / out_v, which point at the FIRST KEPT ROW in the caller's planes.  rows == NULL keeps
 * everything.  Finish with raisr_hip_synchronize(). */
typedef struct raisr_hip_rows {
    int y_skip, y_keep, c_skip, c_keep;
    int stage;      /* 0: upload, kernels and download; 1: upload + kernels only; 2: download only.  Downloads into
                     * pageable memory block the calling thread until the band's kernels are done, so a caller with
                     * several bands issues stage 1 for all of them before the first stage 2. */
} raisr_hip_rows;
int raisr_hip_process_host_async(raisr_hip_ctx *ctx,
                                 const void *in_y, size_t in_y_pitch, void *out_y, size_t out_y_pitch,
                                 const void *in_u, size_t in_u_pitch, void *out_u, size_t out_u_pitch,
                                 const void *in_v, size_t in_v_pitch, void *out_v, size_t out_v_pitch,
                                 int chroma_in_w, int chroma_in_h, int chroma_out_w, int chroma_out_h,
                                 const raisr_hip_rows *rows);

/* Streamed host pipeline ---------------------------------------------------------------------
 * A ring of `depth` contexts: submit() enqueues frame n's upload, kernels and download on lane n % depth and returns;
 * collect() waits for the oldest frame.  With page-locked planes (raisr_hip_host_alloc, or the caller's own buffers
 * through raisr_hip_host_register) frame n+1's upload and frame n-1's download overlap frame n's kernels; pageable planes
 * work too, without the overlap (the runtime stages them on the calling thread).  This is the batch entry beside the
 * synchronous RNLProcess (Library/Raisr.cpp:1294-1397), which keeps its one-frame contract.  One thread drives a stream. */
/* Advanced (what the stream ring is built from): run a context's host-plane entry points on caller-owned streams -- kernels on
 * `compute`, every upload on `upload`, every download on `download` (hipStream_t, all three or none; NULLs restore the context's
 * own streams).  Several contexts sharing the same three streams form a stage-ordered pipeline: uploads, kernels and downloads
 * of consecutive frames each run back to back, in frame order, with device-side events between the stages.
 * raisr_hip_synchronize() then waits for that context's last frame only.  The streams must outlive their use by the context. */
int  raisr_hip_use_streams(raisr_hip_ctx *ctx, void *compute, void *upload, void *download);
int  raisr_hip_set_chunks(raisr_hip_ctx *ctx, int n);               /* host-plane entry: last pass in n row ranges, finished rows downloaded early (1..8) */
int  raisr_hip_set_after(raisr_hip_ctx *ctx, raisr_hip_ctx *prev);   /* bands of one frame: ctx's Y kernels (host-plane entry) start after prev's */

typedef struct raisr_hip_stream raisr_hip_stream;
#define RAISR_HIP_STREAM_MAX_DEPTH 4                                                    /* frames in flight per ring: more never measured faster */
int  raisr_hip_stream_create(raisr_hip_stream **out, int device_index, int depth);     /* depth 1..RAISR_HIP_STREAM_MAX_DEPTH, else RAISR_HIP_EINVAL */
/* The same ring over SEVERAL GPUs from one host thread (north_star: "host code stays C++ ... frames shard across the 8 GPUs";
 * the reference reaches its socket numbers only with N processes, docs/performance.md:8-13): n devices x depth lanes, frame i
 * runs on devices[i % n] (lane (i / n) % depth of that device), collect() returns frames in submission order.  Every device has
 * its own lanes, streams, scratch planes and page-locked bounce memory; raisr_hip_stream_set_model() packs the model ONCE,
 * uploads it to devices[0] and hands it to the others with raisr_hip_broadcast_model_blob_devices().  A device may be listed
 * more than once (tests; two rings' worth of lanes on one GPU).  n = 1..RAISR_HIP_STREAM_MAX_DEVICES, depth as above;
 * raisr_hip_stream_depth() = n * depth frames in flight.  Frame planes must be page-locked memory every listed device can
 * reach (raisr_hip_host_alloc allocates it so) or pageable memory. */
#define RAISR_HIP_STREAM_MAX_DEVICES 16
int  raisr_hip_stream_create_multi(raisr_hip_stream **out, const int *devices, int n, int depth);
int  raisr_hip_stream_device_count(const raisr_hip_stream *s);                         /* n of create_multi (1 for raisr_hip_stream_create) */
int  raisr_hip_stream_device_of_frame(const raisr_hip_stream *s, unsigned long long frame_index);   /* devices[frame_index % n] */
/* "0,2,3" / "all" / "" -> device list (what RAISR_HIP_DEVICES and the plugin's device option carry); returns the count written
 * (<= max), 0 for an empty string, -1 for a malformed one or an index the runtime does not have. */
int  raisr_hip_parse_device_list(const char *text, int *devices, int max);
int  raisr_hip_parse_device_list_n(const char *text, int devices_present, int *devices, int max);   /* the same against a given device count (no runtime call) */
/* The ring's frame order as a pure function: frame_index runs on device slot frame_index % n_devices, on that device's lane
 * (frame_index / n_devices) % depth -- what submit() and collect() walk. */
void raisr_hip_ring_slot(int n_devices, int depth, unsigned long long frame_index, int *device_slot, int *lane_on_device);
void raisr_hip_stream_destroy(raisr_hip_stream *s);
int  raisr_hip_stream_depth(const raisr_hip_stream *s);
int  raisr_hip_stream_set_model(raisr_hip_stream *s, int pass_index, const float *bank, int hashkeys, int pixel_types,
                                const double qstr[2], const double qcoh[2], int quant_angle);
int  raisr_hip_stream_set_model_blob_device(raisr_hip_stream *s, int pass_index, const void *device_blob, size_t bytes,
                                            void *stream);                              /* after raisr_hip_broadcast_model_blob */
int  raisr_hip_stream_set_fast(raisr_hip_stream *s, int level);                        /* raisr_hip_set_fast on every lane; nothing in flight */
int  raisr_hip_stream_configure(raisr_hip_stream *s, const raisr_hip_config *cfg);
int  raisr_hip_stream_set_blending(raisr_hip_stream *s, int blending);                 /* BlendingMode of the frames submitted from now on */
int  raisr_hip_stream_submit(raisr_hip_stream *s,
                             const void *in_y, size_t in_y_pitch, void *out_y, size_t out_y_pitch,
                             const void *in_u, size_t in_u_pitch, void *out_u, size_t out_u_pitch,
                             const void *in_v, size_t in_v_pitch, void *out_v, size_t out_v_pitch,
                             int chroma_in_w, int chroma_in_h, int chroma_out_w, int chroma_out_h);
int  raisr_hip_stream_collect(raisr_hip_stream *s);
int  raisr_hip_stream_in_flight(const raisr_hip_stream *s);
int  raisr_hip_stream_quiesce(raisr_hip_stream *s);                                      /* wait for every frame in flight, collect none */
void *raisr_hip_host_alloc(size_t bytes);            /* page-locked host memory for frame planes */
void raisr_hip_host_free(void *p);
int  raisr_hip_host_register(void *p, size_t bytes); /* page-lock memory the caller already owns; RAISR_HIP_ESTATE: (part of) the range is page-locked already */
int  raisr_hip_host_unregister(void *p);
int  raisr_hip_host_is_page_locked(const void *p); /* 1: p lies in memory from raisr_hip_host_alloc / hipHostMalloc or in a registered range; 0: pageable (or not host memory) */

/* NON-bit-exact fast mode (north_star's MFMA question): DEVELOPMENT BUILDS ONLY since round 4.  The product
 * library refuses every level > 0 (and RAISR_HIP_FAST > 0 at create time) with RAISR_HIP_EINVAL: the matrix-core filter stage
 * measured slower than the exact path (docs/EXPERIMENTS.md), so north_star's question is answered "no".  In a build with
 * -DRAISR_HIP_DEV: buckets exact (level 1) or approximate without the exact re-hash (level 2), the 121-tap dot product of
 * DotProdPatch_AVX512_32f (Raisr_AVX512.cpp:134-149) on the matrix cores with binary16 coefficients; ratio 2, 8/10-bit, asm
 * avx2/avx512 only; bounded by tests/test_gpu_fast_mode.py.  The entry points stay in the ABI so that callers written against
 * rounds 2-3 link. */
int raisr_hip_set_fast(raisr_hip_ctx *ctx, int on);
int raisr_hip_get_fast(const raisr_hip_ctx *ctx);

/* Tracing ------------------------------------------------------------------------------------- */
/* Per-kernel HIP-event timing of subsequent process calls: events are recorded around every kernel on
 * the stream it is launched on.  _read() returns the number of distinct kernels and fills, per kernel,
 * its name (64 bytes each), the summed milliseconds and the launch count since _enable(ctx, 1);
 * the caller synchronises first. */
int raisr_hip_kernel_timing_enable(raisr_hip_ctx *ctx, int on);
int raisr_hip_kernel_timing_read(raisr_hip_ctx *ctx, char *names_out, float *total_ms_out, int *count_out, int max_kernels);
/* (The introspection hooks of the test suite -- raisr_hip_debug_*, raisr_hip_profile_kernels -- are declared in raisr_hip_debug.h
 * and exist in builds with -DRAISR_HIP_TESTHOOKS only: libraisr_hip_testhooks.so, not the product library.) */

#ifdef __cplusplus
}
#endif
#endif /* RAISR_HIP_H */
