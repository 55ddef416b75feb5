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
#if defined(RAISR_HIP_DEV) && !defined(RAISR_HIP_TESTHOOKS)
#define RAISR_HIP_TESTHOOKS 1
#endif
#ifdef RAISR_HIP_TESTHOOKS
#include <atomic>
#include "../../include/raisr_hip_debug.h"
// test-hooks flavour only: fault injection into the multi-device model hand-over (RAISR_HIP_TEST_FAIL_BLOB_SLOT=k: the allocation on
// device slot k "fails") and a count of the per-device staging blobs currently allocated -- tests/test_gpu_stream_multi.py checks that a
// failure on slot k leaves none of slots 0..k-1 behind
static std::atomic<int> g_live_blobs{0};
extern "C" int raisr_hip_debug_stream_live_blobs(void) { return g_live_blobs.load(); }
static bool injected_failure(size_t slot) { const char* e = getenv("RAISR_HIP_TEST_FAIL_BLOB_SLOT"); return e && *e && (size_t)atoi(e) == slot; }
#define RAISR_BLOB_LIVE(delta) g_live_blobs.fetch_add(delta)
#else
static bool injected_failure(size_t) { return false; }
#define RAISR_BLOB_LIVE(delta) ((void)0)
#endif

struct raisr_hip_stream {
    std::vector<int> devices;        // one entry per device slot (a device may appear twice); lane L runs on devices[L % devices.size()]
    std::vector<raisr_hip_ctx*> lanes;
    std::vector<char> busy;          // lane has a submitted, not yet collected frame
    size_t head = 0, tail = 0;       // next lane to submit to / to collect from
    bool configured = false;
    // One stream per lane, at most RAISR_HIP_STREAM_MAX_DEPTH = 4 per device (the runtime has four hardware queues per device;
    // more streams alias onto shared queues): a lane runs whole frames -- uploads, Y kernels, the two cheap chroma upscales, one
    // download -- on its stream.
    // Measured on the way here (scripts/stream_probe.py, stream_trace.sh, copy_engine_probe.hip; 1080p -> 4K yuv420p, PCIe
    // ceiling of the box 4.19 k fps): this layout 3.9-4.2 k fps in 18 of 18 runs; two streams per lane (chroma on its own,
    // round 2a) anywhere between 2.6 k and 4.0 k; six lanes on four streams 2.7 k; uploads on a shared upload stream make the
    // runtime hand the downloads to a copy KERNEL instead of the DMA engine (takes CUs from the next frame's kernels): 2.0-3.0 k.
    std::vector<hipStream_t> comp;   // comp[L]: lane L's stream (on the lane's device), empty with RAISR_HIP_RING_LANE_STREAMS=1
    int device_of_lane(size_t lane) const { return devices[lane % devices.size()]; }
};

static void destroy_streams(raisr_hip_stream* s)
{
    for (size_t i = 0; i < s->comp.size(); i++)
        if (s->comp[i]) { (void)hipSetDevice(s->device_of_lane(i)); (void)hipStreamDestroy(s->comp[i]); }
    s->comp.clear();
}

extern "C" {

// Frame i -> lane i % (n * depth); lanes are laid out device-interleaved (lane L on devices[L % n]), so frame i runs on
// devices[i % n] and consecutive frames of one device use its `depth` lanes in turn.
int raisr_hip_stream_create_multi(raisr_hip_stream** out, const int* devices, int n, int depth)
{
    if (!out || !devices || n < 1 || n > RAISR_HIP_STREAM_MAX_DEVICES || depth < 1 || depth > RAISR_HIP_STREAM_MAX_DEPTH) return RAISR_HIP_EINVAL;   // more lanes per device were never faster: refused, not clamped
    raisr_hip_stream* s = new raisr_hip_stream();
    s->devices.assign(devices, devices + n);
    const size_t nl = (size_t)n * (size_t)depth;
    auto drop = [&](int rc) {
        for (raisr_hip_ctx* p : s->lanes) { (void)raisr_hip_use_streams(p, nullptr, nullptr, nullptr); raisr_hip_destroy(p); }
        destroy_streams(s);
        delete s;
        return rc;
    };
    for (size_t i = 0; i < nl; i++) {
        raisr_hip_ctx* c = nullptr;
        const int rc = raisr_hip_create(&c, s->device_of_lane(i));
        if (rc != RAISR_HIP_OK) return drop(rc);
        s->lanes.push_back(c);
    }
    s->busy.assign(nl, 0);
    const char* legacy = getenv("RAISR_HIP_RING_LANE_STREAMS");         // A/B switch: 1 = every lane on its own streams (round-2a behaviour)
    if (!(legacy && atoi(legacy) != 0)) {
        s->comp.assign(nl, nullptr);
        bool ok = true;
        for (size_t i = 0; ok && i < nl; i++)
            ok = hipSetDevice(s->device_of_lane(i)) == hipSuccess && hipStreamCreateWithFlags(&s->comp[i], hipStreamNonBlocking) == hipSuccess &&
                 raisr_hip_use_streams(s->lanes[i], s->comp[i], s->comp[i], s->comp[i]) == RAISR_HIP_OK;
        if (!ok) return drop(RAISR_HIP_ERUNTIME);
    }
    *out = s;
    return RAISR_HIP_OK;
}

int raisr_hip_stream_create(raisr_hip_stream** out, int device_index, int depth)
{
    return raisr_hip_stream_create_multi(out, &device_index, 1, depth);
}

int raisr_hip_stream_device_count(const raisr_hip_stream* s) { return s ? (int)s->devices.size() : 0; }

int raisr_hip_stream_device_of_frame(const raisr_hip_stream* s, unsigned long long frame_index)
{
    return (s && !s->devices.empty()) ? s->devices[(size_t)(frame_index % s->devices.size())] : -1;
}

int raisr_hip_parse_device_list(const char* text, int* devices, int max)
{
    return raisr_hip_parse_device_list_n(text, raisr_hip_device_count(), devices, max);
}

// frame -> (device slot, lane on that device) of an n-device ring with `depth` lanes per device: the map submit() and collect() walk
void raisr_hip_ring_slot(int n_devices, int depth, unsigned long long frame_index, int* device_slot, int* lane_on_device)
{
    if (n_devices < 1) n_devices = 1;
    if (depth < 1) depth = 1;
    const unsigned long long lane = frame_index % ((unsigned long long)n_devices * (unsigned long long)depth);
    if (device_slot) *device_slot = (int)(lane % (unsigned long long)n_devices);
    if (lane_on_device) *lane_on_device = (int)(lane / (unsigned long long)n_devices);
}

int raisr_hip_parse_device_list_n(const char* text, int have, int* devices, int max)
{
    if (!text || !devices || max < 1) return -1;
    while (*text == ' ') text++;
    if (!*text) return 0;
    if (!strcmp(text, "all")) {
        int n = 0;
        for (int d = 0; d < have && n < max; d++) devices[n++] = d;
        return n > 0 ? n : -1;
    }
    int n = 0;
    const char* p = text;
    while (*p) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v >= have || n >= max) return -1;
        devices[n++] = (int)v;
        p = end;
        while (*p == ' ') p++;
        if (*p == ',') { p++; if (!*p) return -1; } else if (*p) return -1;
    }
    return n;
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
    const size_t nd = s->devices.size();
    if (nd == 1) {
        for (raisr_hip_ctx* c : s->lanes) {
            const int rc = raisr_hip_set_model(c, pass_index, bank, hashkeys, pixel_types, qstr, qcoh, quant_angle);
            if (rc != RAISR_HIP_OK) return rc;
        }
        return RAISR_HIP_OK;
    }
    // several devices: the model is packed once, crosses PCIe once (to devices[0]) and reaches the other devices by the
    // in-process broadcast (RCCL over xGMI); every lane then takes a device-local copy
    const size_t bytes = raisr_hip_model_blob_bytes(hashkeys, pixel_types);
    if (!bytes) return RAISR_HIP_EINVAL;
    std::vector<unsigned char> host(bytes);
    int rc = raisr_hip_pack_model_blob(host.data(), bank, hashkeys, pixel_types, qstr, qcoh, quant_angle);
    if (rc != RAISR_HIP_OK) return rc;
    std::vector<void*> blobs(nd, nullptr);
    auto release = [&]() { for (size_t d = 0; d < nd; d++) if (blobs[d]) { (void)hipSetDevice(s->devices[d]); (void)hipFree(blobs[d]); blobs[d] = nullptr; RAISR_BLOB_LIVE(-1); } };
    for (size_t d = 0; d < nd && rc == RAISR_HIP_OK; d++) {
        if (injected_failure(d) || hipSetDevice(s->devices[d]) != hipSuccess || hipMalloc(&blobs[d], bytes) != hipSuccess) { (void)hipGetLastError(); blobs[d] = nullptr; rc = RAISR_HIP_ENOMEM; }
        else RAISR_BLOB_LIVE(1);
    }
    if (rc == RAISR_HIP_OK && (hipSetDevice(s->devices[0]) != hipSuccess || hipMemcpy(blobs[0], host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = RAISR_HIP_ERUNTIME;
    if (rc == RAISR_HIP_OK) rc = raisr_hip_broadcast_model_blob_devices(s->devices.data(), (int)nd, blobs.data(), bytes);
    for (size_t i = 0; i < s->lanes.size() && rc == RAISR_HIP_OK; i++)
        rc = raisr_hip_set_model_blob_device(s->lanes[i], pass_index, blobs[i % nd], bytes, nullptr);
    release();
    return rc;
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
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;        // portable: every device of a multi-device ring copies from / to it
    return p;
}
void raisr_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }
int raisr_hip_host_register(void* p, size_t bytes)
{
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);          // portable: every device of a multi-device ring
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
