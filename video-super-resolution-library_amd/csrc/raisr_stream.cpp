// raisr_stream.cpp -- streamed host pipeline over the device ABI: a ring of `depth` contexts (one HIP stream-ordered lane
// each), so that frame n+1's upload and frame n-1's download overlap frame n's kernels.  This is the batch / async entry
// SURVEY.md s7 step 6 asks for; RNLHandler_Process keeps its synchronous one-frame contract (reference
// Library/Raisr.cpp:1294-1397) and is untouched.
//
// Overlap needs page-locked host planes: a copy from pageable memory is staged by the runtime on the calling thread.
// The library therefore also exports raisr_hip_host_alloc / raisr_hip_host_register so that a host (FFmpeg: a custom
// get_video_buffer pool) can hand over planes the copy engines read and write directly.  With pageable planes the
// ring still works, without overlap.
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/raisr_hip.h"

struct raisr_hip_stream {
    int device = 0;
    std::vector<raisr_hip_ctx*> lanes;
    std::vector<char> busy;          // lane has a submitted, not yet collected frame
    size_t head = 0, tail = 0;       // next lane to submit to / to collect from
    bool configured = false;
    // One stream per lane, at most kMaxLanes = 4 of them (the runtime has four hardware queues; more streams alias onto shared
    // queues): a lane runs whole frames -- uploads, Y kernels, the two cheap chroma upscales, one download -- on its stream.
    // Measured on the way here (scripts/stream_probe.py, stream_trace.sh, copy_engine_probe.hip; 1080p -> 4K yuv420p, PCIe
    // ceiling of the box 4.19 k fps): this layout 3.9-4.2 k fps in 18 of 18 runs; two streams per lane (chroma on its own,
    // round 2a) anywhere between 2.6 k and 4.0 k; six lanes on four streams 2.7 k; uploads on a shared upload stream make the
    // runtime hand the downloads to a copy KERNEL instead of the DMA engine (takes CUs from the next frame's kernels): 2.0-3.0 k.
    hipStream_t comp[4] = {nullptr, nullptr, nullptr, nullptr};
    int nstreams = 0;
};

static void destroy_streams(raisr_hip_stream* s)
{
    (void)hipSetDevice(s->device);
    for (hipStream_t& c : s->comp) if (c) (void)hipStreamDestroy(c);
}

extern "C" {

int raisr_hip_stream_create(raisr_hip_stream** out, int device_index, int depth)
{
    if (!out || depth < 1 || depth > RAISR_HIP_STREAM_MAX_DEPTH) return RAISR_HIP_EINVAL;   // more lanes than that were never faster (see raisr_hip_stream::comp): refused, not clamped
    raisr_hip_stream* s = new raisr_hip_stream();
    s->device = device_index;
    for (int i = 0; i < depth; i++) {
        raisr_hip_ctx* c = nullptr;
        const int rc = raisr_hip_create(&c, device_index);
        if (rc != RAISR_HIP_OK) {
            for (raisr_hip_ctx* p : s->lanes) raisr_hip_destroy(p);
            delete s;
            return rc;
        }
        s->lanes.push_back(c);
    }
    s->busy.assign((size_t)depth, 0);
    const char* legacy = getenv("RAISR_HIP_RING_LANE_STREAMS");         // A/B switch: 1 = every lane on its own streams (round-2a behaviour)
    if (!(legacy && atoi(legacy) != 0)) {
        s->nstreams = depth;
        bool ok = hipSetDevice(device_index) == hipSuccess;
        for (int i = 0; ok && i < s->nstreams; i++) ok = hipStreamCreateWithFlags(&s->comp[i], hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < depth; i++) {
            hipStream_t c = s->comp[i % s->nstreams];
            ok = raisr_hip_use_streams(s->lanes[(size_t)i], c, c, c) == RAISR_HIP_OK;
        }
        if (!ok) {
            for (raisr_hip_ctx* p : s->lanes) raisr_hip_destroy(p);
            destroy_streams(s);
            delete s;
            return RAISR_HIP_ERUNTIME;
        }
    }
    *out = s;
    return RAISR_HIP_OK;
}

void raisr_hip_stream_destroy(raisr_hip_stream* s)
{
    if (!s) return;
    for (raisr_hip_ctx* c : s->lanes) { (void)raisr_hip_synchronize(c); (void)raisr_hip_use_streams(c, nullptr, nullptr, nullptr); raisr_hip_destroy(c); }
    destroy_streams(s);
    delete s;
}

int raisr_hip_stream_depth(const raisr_hip_stream* s) { return s ? (int)s->lanes.size() : 0; }

int raisr_hip_stream_set_model(raisr_hip_stream* s, int pass_index, const float* bank, int hashkeys, int pixel_types,
                               const double qstr[2], const double qcoh[2], int quant_angle)
{
    if (!s) return RAISR_HIP_EINVAL;
    for (size_t i = 0; i < s->lanes.size(); i++)
        if (s->busy[i]) return RAISR_HIP_ESTATE;                   // a lane's kernels may still be reading its bank
    for (raisr_hip_ctx* c : s->lanes) {
        const int rc = raisr_hip_set_model(c, pass_index, bank, hashkeys, pixel_types, qstr, qcoh, quant_angle);
        if (rc != RAISR_HIP_OK) return rc;
    }
    return RAISR_HIP_OK;
}

// Multi-GPU streams (BASELINE C5): the rank received the packed blob by raisr_hip_broadcast_model_blob; every lane copies it.
int raisr_hip_stream_set_model_blob_device(raisr_hip_stream* s, int pass_index, const void* device_blob, size_t bytes, void* stream)
{
    if (!s) return RAISR_HIP_EINVAL;
    for (size_t i = 0; i < s->lanes.size(); i++)
        if (s->busy[i]) return RAISR_HIP_ESTATE;
    for (raisr_hip_ctx* c : s->lanes) {
        const int rc = raisr_hip_set_model_blob_device(c, pass_index, device_blob, bytes, stream);
        if (rc != RAISR_HIP_OK) return rc;
    }
    return RAISR_HIP_OK;
}

int raisr_hip_stream_set_fast(raisr_hip_stream* s, int level)
{
    if (!s) return RAISR_HIP_EINVAL;
    for (size_t i = 0; i < s->lanes.size(); i++)
        if (s->busy[i]) return RAISR_HIP_ESTATE;
    for (raisr_hip_ctx* c : s->lanes) {
        const int rc = raisr_hip_set_fast(c, level);
        if (rc != RAISR_HIP_OK) return rc;
    }
    return RAISR_HIP_OK;
}

// BlendingMode of the frames submitted from now on (a frame's mode is read when its kernels are enqueued)
int raisr_hip_stream_set_blending(raisr_hip_stream* s, int blending)
{
    if (!s) return RAISR_HIP_EINVAL;
    for (raisr_hip_ctx* c : s->lanes) {
        const int rc = raisr_hip_set_blending(c, blending);
        if (rc != RAISR_HIP_OK) return rc;
    }
    return RAISR_HIP_OK;
}

int raisr_hip_stream_configure(raisr_hip_stream* s, const raisr_hip_config* cfg)
{
    if (!s || !cfg) return RAISR_HIP_EINVAL;
    for (size_t i = 0; i < s->lanes.size(); i++)
        if (s->busy[i]) return RAISR_HIP_ESTATE;                   // collect everything before changing the geometry
    s->configured = false;                                         // lanes in mixed geometry are never submitted to: all of them or none
    for (raisr_hip_ctx* c : s->lanes) {
        const int rc = raisr_hip_configure(c, cfg);
        if (rc != RAISR_HIP_OK) return rc;
    }
    s->configured = true;
    return RAISR_HIP_OK;
}

// Enqueue one frame (upload, kernels, download -- all asynchronous when the planes are page-locked).  Returns
// RAISR_HIP_ESTATE when `depth` frames are already in flight: collect first.  The planes must stay valid and untouched
// until the matching collect returns.
int raisr_hip_stream_submit(raisr_hip_stream* s,
                            const void* in_y, size_t in_y_pitch, void* out_y, size_t out_y_pitch,
                            const void* in_u, size_t in_u_pitch, void* out_u, size_t out_u_pitch,
                            const void* in_v, size_t in_v_pitch, void* out_v, size_t out_v_pitch,
                            int cin_w, int cin_h, int cout_w, int cout_h)
{
    if (!s || !s->configured) return RAISR_HIP_ESTATE;
    const size_t n = s->lanes.size(), lane = s->head % n;
    if (s->busy[lane]) return RAISR_HIP_ESTATE;
    const int rc = raisr_hip_process_host_async(s->lanes[lane], in_y, in_y_pitch, out_y, out_y_pitch, in_u, in_u_pitch, out_u, out_u_pitch,
                                                in_v, in_v_pitch, out_v, out_v_pitch, cin_w, cin_h, cout_w, cout_h, nullptr);
    if (rc != RAISR_HIP_OK) return rc;
    s->busy[lane] = 1;
    s->head++;
    return RAISR_HIP_OK;
}

// Wait for the OLDEST submitted frame; its output planes are complete when this returns.  RAISR_HIP_ESTATE if nothing is in flight.
int raisr_hip_stream_collect(raisr_hip_stream* s)
{
    if (!s) return RAISR_HIP_EINVAL;
    const size_t n = s->lanes.size(), lane = s->tail % n;
    if (!s->busy[lane]) return RAISR_HIP_ESTATE;
    const int rc = raisr_hip_synchronize(s->lanes[lane]);
    s->busy[lane] = 0;
    s->tail++;
    return rc;
}

int raisr_hip_stream_in_flight(const raisr_hip_stream* s) { return s ? (int)(s->head - s->tail) : 0; }

// Wait until every submitted frame is complete WITHOUT collecting any (the frames stay in flight for the caller; their collects
// return at once): what a host does before it unpins or frees memory the frames in flight may still be copied from or to.
int raisr_hip_stream_quiesce(raisr_hip_stream* s)
{
    if (!s) return RAISR_HIP_EINVAL;
    int rc = RAISR_HIP_OK;
    for (size_t i = 0; i < s->lanes.size(); i++)
        if (s->busy[i]) { const int r = raisr_hip_synchronize(s->lanes[i]); if (rc == RAISR_HIP_OK) rc = r; }
    return rc;
}

// Page-locked host memory for frame planes (what makes the copies of a stream asynchronous).
void* raisr_hip_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void raisr_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }
int raisr_hip_host_register(void* p, size_t bytes)
{
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipSuccess) return RAISR_HIP_OK;
    (void)hipGetLastError();                                                     // a refused registration is not a sticky error
    return e == hipErrorHostMemoryAlreadyRegistered ? RAISR_HIP_ESTATE : RAISR_HIP_ERUNTIME;
}
int raisr_hip_host_unregister(void* p) { return hipHostUnregister(p) == hipSuccess ? RAISR_HIP_OK : RAISR_HIP_ERUNTIME; }
int raisr_hip_host_is_page_locked(const void* p)
{
    if (!p) return 0;
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }     // ordinary memory: not an error of the caller's
    return a.type == hipMemoryTypeHost ? 1 : 0;
}

}  // extern "C"
