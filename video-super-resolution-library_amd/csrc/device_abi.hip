// device_abi.hip -- the C ABI of include/raisr_hip.h (contexts, models, geometry, frame entry points, stream plumbing) and the
// ONE translation unit the gfx950 kernels are compiled in.  The kernels live in the kernels_*.h headers next to this file:
//
//   kernels_resize.h        k_resize / k_resize2x / k_copy   cheap upscale (stand-in for ippiResizeLinear, Raisr.cpp:947-958)  in -> LR
//   kernels_hash.h          hash_px_*, hash_phase, k_hash    the reference's tensor + hash (Raisr_AVX512.cpp:69-131,175-258;
//                                                            tail columns also Raisr_AVX256.cpp:393-472)
//   kernels_hash_certify.h  approx_hash, hash_phase_ac       certified hashing: approximate tensor, rigorous bounds, exact fallback
//   kernels_filter.h        filter_phase, k_filter,          hash-indexed 121-tap filter + accept test (Raisr_AVX512.cpp:134-149,
//                           k_hashfilter, k_hashfilter_ac    Raisr.cpp:1196-1200); k_hashfilter_ac = PRODUCTION kernel of the fp32 numerics:
//                                                            per 64 x 16 tile, hash stage and filter stage sharing one LR window in LDS
//   kernels_fp16.h          k_hashfilter16, k_blend16, ...   the AVX512-FP16 numerics in binary16 (Raisr_AVX512FP16.cpp)
//   kernels_blend.h         k_blend, k_blend_rand            census-transform blend, clamp, narrow, borders (Raisr_AVX256.cpp:68-166,
//                                                            Raisr.cpp:999-1028,1252-1265)
// builds with -DRAISR_HIP_TESTHOOKS (libraisr_hip_testhooks.so: the test suite's flavour) add the hooks of include/raisr_hip_debug.h and
//   kernels_split.h         k_hash_ac, k_fix_*, k_filter_lds16   comparison pipeline: un-fused, filter bank in LDS (RAISR_HIP_SPLIT=1)
//   kernels_fix.h           k_fix_ac                         comparison pipeline: exact path of the uncertified pixels as its own kernel (RAISR_HIP_DEFER=1)
// development builds (-DRAISR_HIP_DEV) furthermore
//   kernels_fast.h          k_filter_mfma                    NOT bit-exact: filter stage on the matrix cores (measured slower: rejected)
//   kernels_probes.h, kernels_experiments.h                  timing probes (output wrong) and rejected experiments (docs/EXPERIMENTS.md)
//
// Pipeline per RAISR pass (whole-frame semantics of the reference's processSegment(), Library/Raisr.cpp:890-1289, run with
// threadcount=1):  k_resize -> k_hashfilter_ac -> k_blend.
//
// Numeric contract: every floating-point operation in the kernels maps to exactly one IEEE-754 binary32 (binary16) operation of
// the cited reference lines ("strict source" semantics).  This file MUST be built with -ffp-contract=off and without fast-math;
// FMAs appear only where the reference has an explicit fmadd intrinsic.
//
// Design notes (MI355X): ~1.3 kFLOP per output pixel per pass against ~1.25 compulsory HBM bytes: not an HBM-bound path.  The main
// kernel is instruction-issue bound at the 16 waves per CU its LDS and registers allow, with VALU, LDS and vector L1 each 40-60 % busy,
// so the kernels are organised around instruction count and VALU / LDS / L1 efficiency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <map>
#include <vector>

#include "../../include/raisr_hip.h"
#if defined(RAISR_HIP_DEV) && !defined(RAISR_HIP_TESTHOOKS)
#define RAISR_HIP_TESTHOOKS 1         /* development builds carry the test hooks as well */
#endif
#ifdef RAISR_HIP_TESTHOOKS
#include "../../include/raisr_hip_debug.h"
#endif
#include "host_copy.h"               // RowCopyPool, HostBounce: pageable host planes go through page-locked bounce memory
#include "x86_approx_tables.h"
#include "x86_approx_dev.h"
#include "x86_fp16_tables.h"

#if defined(__FAST_MATH__)
#error "device_abi.hip must be compiled without fast-math"
#endif

typedef float f2 __attribute__((ext_vector_type(2)));


namespace {

#include "kernels_common.h"          // PassParams, xcd_tile, stage_tile
#include "kernels_resize.h"          // k_resize, k_resize2x, k_copy
#include "kernels_hash.h"            // hash_px_*, hash_phase, k_hash
#include "kernels_hash_certify.h"    // approx_hash, tensor_ac, hash_phase_ac
#include "kernels_fp16.h"            // binary16 pipeline: k_hashfilter16, k_hash16, k_filter16, k_blend16
#ifdef RAISR_HIP_DEV
#include "kernels_probes.h"          // timing probes of the filter stage (output wrong): hooks that kernels_filter.h expands
#endif
#include "kernels_filter.h"          // filter_phase, k_filter, k_hashfilter, k_hashfilter_ac
#ifdef RAISR_HIP_TESTHOOKS           /* comparison pipelines (RAISR_HIP_DEFER, RAISR_HIP_SPLIT): test-hooks builds only */
#include "kernels_fix.h"             // k_fix_ac: deferred exact path of k_hashfilter_ac
#endif
#ifdef RAISR_HIP_DEV
#include "kernels_experiments.h"     // persistent-grid variant of k_hashfilter_ac (-DRAISR_EXP_PERSIST)
#endif
#ifdef RAISR_HIP_TESTHOOKS
#include "kernels_split.h"           // k_hash_ac, k_fix_*, k_filter_lds16
#endif
#ifdef RAISR_HIP_DEV                 /* development builds only: measured slower than the exact path (docs/EXPERIMENTS.md) */
#include "kernels_fast.h"            // k_filter_mfma
#endif
#include "kernels_blend.h"           // k_blend, k_blend_rand

// ------------------------------------------------------------------------------------------------
// Host side of the C ABI
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;
// The matrix-core filter stage (north_star's MFMA question) is answered "no": with exact buckets it equals the exact path's speed,
// and since round 4 it is slower (docs/EXPERIMENTS.md).  It stays in development builds as a measurement.
[[maybe_unused]] const char* const kNoFastMode = "the NON-bit-exact fast mode (matrix-core filter stage) is not part of the product build: it is slower than the "
                                "exact path; a development build (-DRAISR_HIP_DEV) keeps it for measurements";

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    g_err = what;
    if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
    return code;
}

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return fail(RAISR_HIP_ERUNTIME, #expr, _e);      \
    } while (0)

struct BlobHeader {
    uint32_t magic;
    int32_t hashkeys, pixel_types, quant_angle;
    float qangle, qstr[2], qcoh[2];
    uint16_t qangle16, qstr16[2], qcoh16[2];   // binary16 flavours for the AVX512-FP16 pipeline
    uint16_t pad16;
    uint32_t pad[4];
};
static_assert(sizeof(BlobHeader) == kBlobHeader, "blob header size");
constexpr uint32_t kBlobMagic = 0x52534152u;   // "RASR"

// blob = header | fp32 bank [rows][128] | fp16 bank [rows][4][16] half2
inline size_t blob_f32_bytes(int rows) { return (size_t)rows * kTapsPad * sizeof(float); }
inline size_t blob_f16_bytes(int rows) { return (size_t)rows * 64 * sizeof(uint32_t); }

struct ModelDev {
    void* blob = nullptr;
    size_t bytes = 0;
    BlobHeader h{};
    bool valid = false;
    int zero_bucket[2] = {0, 0};           // bucket of the all-zero tensor, AVX-512 / AVX2 flavour (from the device hash code)
    _Float16* bank_mfma = nullptr;         // fast mode: binary16 B panels of k_filter_mfma, built on first use
    bool bank_mfma_valid = false;
    // symmetric filter stage (kernels_filter.h, filter_phase<.., SYM>): which bank rows are NOT palindromes
    float* bank_lm = nullptr;              // lane-major copy of the blob's fp32 bank (k_lane_major_bank): what the filter stage reads
    size_t bank_lm_bytes = 0;
    uint32_t* d_asym = nullptr;            // [32] bitmap over bucket * pixel_types + type (allocated with the first model)
    int asym_rows = -1;                    // number of set bits; -1 = not scanned (the symmetric kernel is never chosen then)
    // binary16 pipeline: thresholds folded onto the dividends of the hash's strength and coherence divisions (Pass16)
    uint16_t ls16[2] = {0x7e00, 0x7e00};
    float cm16[2] = {0.f, 0.f};
    int ct16[2] = {0, 0};
    int fold16 = 0;
};

struct KernelTimer {
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;                 // recs[i] uses pool[2i], pool[2i+1]
    std::vector<hipEvent_t> pool;          // created once at enable time, outside any timed region
    std::vector<std::string> names;
    bool enabled = false;
    size_t cap = 4096;                     // launches recorded per enable; later launches run untimed
};

int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// Gaussian weights: gGaussian2D{8,10,16}bit (Raisr_globals.h:208-264) = (float)((double)NF * literal);
// the literal table is symmetric, Q is its upper-left quadrant.
const double kGaussQ[6][6] = {
    {7.76554e-05, 0.000239195, 0.0005738, 0.001072, 0.00155975, 0.00176743},
    {0.000239195, 0.000736774, 0.00176743, 0.00330199, 0.00480437, 0.00544406},
    {0.0005738, 0.00176743, 0.00423984, 0.00792107, 0.0115251, 0.0130596},
    {0.001072, 0.00330199, 0.00792107, 0.0147985, 0.0215317, 0.0243986},
    {0.00155975, 0.00480437, 0.0115251, 0.0215317, 0.0313284, 0.0354998},
    {0.00176743, 0.00544406, 0.0130596, 0.0243986, 0.0354998, 0.0402265},
};

GaussW make_gauss(int bits)
{
    GaussW g{};
    const float maxv = bits == 8 ? 255.0f : (bits == 10 ? 1023.0f : 65535.0f);
    volatile float nf = 1.0f / (maxv * maxv * 2.0f * 2.0f);
    for (int i = 0; i < 11; i++)
        for (int k = 0; k < 11; k++) {
            const int qi = i < 6 ? i : 10 - i, qk = k < 6 ? k : 10 - k;
            g.wT[k][i] = (float)((double)nf * kGaussQ[qi][qk]);
        }
    return g;
}

// Separable weights and error constants of the certified hash stage (see the comment above approx_hash).
// us_i = sqrt(NF * literal_ii): the literal table is the outer product of a 1-D Gaussian up to its 6-digit truncation.
// eps_w is measured on the very fp32 constants the two paths use, so it is a bound, not an estimate.
SepW make_sep(const GaussW& g)
{
    SepW S{};
    for (int i = 0; i < 11; i++) S.us[i] = (float)sqrt((double)g.wT[i][i]);
    double eps_w = 0.0;
    for (int i = 0; i < 11; i++)
        for (int k = 0; k < 11; k++) {
            const double r = (double)S.us[i] * (double)S.us[k] / (double)g.wT[k][i] - 1.0;
            if (fabs(r) > eps_w) eps_w = fabs(r);
        }
    const double u = 5.9604644775390625e-8;                     // 2^-24
    const double eps = 1.05 * (eps_w + 48.0 * u);
    const double E[2] = {1.0e-4, 6.5e-4};                       // sup |sqrt14(x)/sqrt(x) - 1|: VRCP14(VRSQRT14), RCPPS(RSQRTPS)
    S.es1 = (float)(1.42 * eps);
    S.es2 = (float)(2e-7 + eps * eps);
    S.eEL = (float)(0.5 * eps + 6.0 * u);
    S.eEb = (float)(0.5 * eps);
    for (int f = 0; f < 2; f++) { S.e105[f] = (float)(1.05 * E[f]); S.e24[f] = (float)(2.4 * E[f]); }
    S.c1e = (float)(1.05 * eps);
    return S;
}

// Column plan of the reference's chunk driver (Raisr.cpp:1065-1066,1246-1250).
void column_plan(int W, int hash_variant, PassParams& P)
{
    const int unroll = hash_variant == RAISR_HIP_HASH_AVX512 ? 16 : (hash_variant == RAISR_HIP_HASH_FP16 ? 32 : 8);
    int loopItr = unroll, c = kMargin;
    P.a_begin = P.a_end = P.b_begin = P.b_end = kMargin;
    bool a_any = false, b_any = false;
    while (c + loopItr <= W - kMargin) {
        if (loopItr >= 16) { if (!a_any) { P.a_begin = c; a_any = true; } P.a_end = c + loopItr; }
        else { if (!b_any) { P.b_begin = c; b_any = true; } P.b_end = c + 8; }
        if (loopItr > 8 && c + 2 * unroll > W - kMargin) loopItr = 8;
        c += loopItr;
    }
    P.c_final = a_any || b_any ? (b_any ? P.b_end : P.a_end) : kMargin;
    if (a_any && b_any && P.a_end > P.b_end) P.c_final = P.a_end;
    // columns inside both ranges are hashed twice (at most 16 of them: one 16-wide chunk)
    P.ov_begin = P.ov_end = 0;
    if (a_any && b_any && P.b_begin < P.a_end) { P.ov_begin = P.b_begin; P.ov_end = P.a_end < P.b_end ? P.a_end : P.b_end; }
}

}  // namespace

struct raisr_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;             // chroma lane of raisr_hip_process_host (overlaps the Y path)
    // stream ring (raisr_hip_use_streams): caller-owned compute / upload / download streams shared by several contexts, so
    // that a streamed job is a stage-ordered pipeline (all uploads in frame order on one stream, all downloads on another)
    hipStream_t up = nullptr, down = nullptr;
    hipStream_t own_stream = nullptr;          // the pooled stream `stream` replaced while a caller-owned one is in use
    hipEvent_t ev_up = nullptr, ev_comp = nullptr, ev_done = nullptr;
    bool done_pending = false;
    bool legacy_pending = false;               // ring context that took the two-stream branch of process_host_async (rows != NULL)
    // bands of one frame on several contexts (raisr_hip_set_after): this context's Y kernels start when the previous band's are
    // done, so that the bands' kernels run one after the other while band k's download overlaps band k+1's kernels
    // host-plane entry, whole frames: the last pass runs in `chunks` row ranges and every finished range is downloaded on the
    // second stream while the next one is computed (RAISR_HIP_CHUNKS; 1 = one download after the frame)
    int chunks = 1;
    hipEvent_t ev_chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    raisr_hip_ctx* after = nullptr;
    hipEvent_t ev_kern = nullptr;
    bool ev_kern_valid = false;
    bool fused = true;                         // one k_hashfilter launch per pass instead of k_hash + k_filter (RAISR_HIP_FUSED=0)
    bool fold16 = true;                        // binary16 hash: strength / coherence thresholds folded onto the dividends (RAISR_HIP_FOLD16=0 keeps the divisions)
    bool sym = true;                           // symmetric filter stage for banks whose rows are (nearly all) palindromes; RAISR_HIP_SYM=0 keeps the eight-load stage
    int blend_rows = 8;                        // census blend: rows per wave of k_blend4 / k_blend4_16 (4, 8, 16), 0 = always the 64 x 16 LDS-tile kernels (RAISR_HIP_BLEND_ROWS: A/B switch)
    int sym_max_rows = 16;                     // ... chosen when at most this many rows are not (RAISR_HIP_SYM_MAX_ROWS); their pixels are redone with eight loads
    bool certify = true;                       // certified hash stage (k_hashfilter_ac / k_hash_ac); RAISR_HIP_CERTIFY=0 keeps the all-exact kernels
    bool defer = false;                        // RAISR_HIP_DEFER=1: uncertified pixels go to a per-frame list and k_fix_ac instead of the in-tile worklist
                                               // (bit-exact; measured slower on every configuration, docs/EXPERIMENTS.md: k_fix_ac costs more than the main kernel gains)
#ifdef RAISR_HIP_TESTHOOKS
    FixAc fixac{};                             // ... the list (sized at configure: tiles of the largest pass x batch depth)
#endif
    bool split = false;                        // certified hash stage and filter stage as separate launches (RAISR_HIP_SPLIT=1)
    int fast = 0;                              // NON-bit-exact fast mode (raisr_hip_set_fast / RAISR_HIP_FAST): 1 = exact buckets, filter stage on the matrix cores; 2 = also keeps the approximate tensor's bucket where it is not certified
    bool lds_filter = true;                    // split pipeline: filter stage with the bank in LDS (RAISR_HIP_LDS_FILTER=0: k_filter)
    int n_cus = 256;                           // persistent k_filter_lds16 grid: one workgroup per CU (multiple of 4)
    float* d_gauss = nullptr;                  // GaussW::wT on the device (16-lane exact tensor of the worklist)
#ifdef RAISR_HIP_TESTHOOKS
    FixLists fix{};                            // split pipeline: worklists of the certified hash stage (sized at configure)
    size_t fix_tiles = 0;
#endif
    int cert_check = 0;                        // tests: every pixel also takes the exact path, certified buckets are compared
    unsigned* d_cert_stats = nullptr;          // 8 counters {uncertain, certified-but-wrong, zone pixels, tiles with a list, overflowed tiles, tiles, flat tiles, -}, accumulated while non-null
    SepW sep{};
    int keep_hash_plane = 0;                   // fused kernel also writes the hash plane (set by raisr_hip_debug_read_stage users)
    raisr_hip_config cfg{};
    int blending = RAISR_HIP_BLEND_COUNT;      // per-call BlendingMode (RNLProcess argument)
    bool configured = false;
    ModelDev model[2];
    // shared small tables
    uint2* d_tab14 = nullptr;
    uint8_t* d_c1tab = nullptr;                // class-1 sign table (k_build_c1tab), 64 KB; null when RAISR_HIP_C1=0
    uint16_t* d_lut = nullptr;
    uint16_t* d_tab16 = nullptr;                // rcpph T, rsqrtph T0, T1
    GaussW16 gauss16{};
    // scratch planes
    void* d_lr[2] = {nullptr, nullptr};         // LR plane per pass, sample type (u8 for 8-bit content, else u16)
    uint8_t* d_hash[2] = {nullptr, nullptr};    // first hash per pixel
    uint8_t* d_hash2[2] = {nullptr, nullptr};   // [H][16] second hash of the tail (overlap) columns
    float* d_hr[2] = {nullptr, nullptr};
    void* d_mid = nullptr;                      // two-pass intermediate (sample type), only when the passes differ in size
    const void* lr0_alias = nullptr;            // set per frame: pass 1 runs at input size on a tightly pitched input plane -> no copy into d_lr[0]
    int passW[2] = {0, 0}, passH[2] = {0, 0};
    // frame batches (raisr_hip_process_y_device_batch): the scratch planes exist batch_cap times, back to back; zb_* describe the
    // batch of the call in progress (1 / 0 / 0 outside one)
    const void* final_out = nullptr;              // the caller's output plane of the frame in progress: the pass that writes it applies sample_shift
    int sample_shift = 0;                         // device-frame entries: samples are MSB-aligned by this many bits (raisr_hip_set_sample_shift)
    int batch_cap = 1;
    int zb_n = 1;
    size_t zb_in_stride = 0, zb_out_stride = 0;      // elements between consecutive caller planes
    GaussW gauss{};
    // device staging for raisr_hip_process_host
    void* d_stage = nullptr; size_t d_stage_bytes = 0;
    HostBounce bounce;                          // page-locked bounce memory for pageable host planes (host_copy.h)
    int numa_node = -1;                         // NUMA node of the device's PCI function (-1: unknown / single node): which row-copy pool serves this context
    hipEvent_t ev_chroma = nullptr;             // chroma lane done (packed-frame download waits for it)
    KernelTimer timer;
};

namespace {

void timer_begin(raisr_hip_ctx* c, const char* name, hipStream_t s, int& slot)
{
    slot = -1;
    KernelTimer& T = c->timer;
    if (!T.enabled || T.recs.size() >= T.cap) return;
    int id = -1;
    for (size_t i = 0; i < T.names.size(); i++) if (T.names[i] == name) { id = (int)i; break; }
    if (id < 0) { T.names.push_back(name); id = (int)T.names.size() - 1; }
    const size_t i = T.recs.size();
    if (2 * i + 1 >= T.pool.size()) return;
    KernelTimer::Rec r{id, T.pool[2 * i], T.pool[2 * i + 1]};
    (void)hipEventRecord(r.a, s);
    T.recs.push_back(r);
    slot = (int)T.recs.size() - 1;
}
void timer_end(raisr_hip_ctx* c, hipStream_t s, int slot)
{
    if (slot >= 0) (void)hipEventRecord(c->timer.recs[slot].b, s);
}

ResizeParams make_resize(int sw, int sh, int spitch, int dw, int dh, int dpitch, int tie)
{
    ResizeParams R{};
    R.sw = sw; R.sh = sh; R.dw = dw; R.dh = dh; R.spitch = spitch; R.dpitch = dpitch;
    R.sstep = R.dstep = 1;
    const int gx = gcd_int(sw, dw), gy = gcd_int(sh, dh);
    R.Sx = sw / gx; R.Dx = dw / gx; R.Sy = sh / gy; R.Dy = dh / gy;
    R.tie_even = tie == RAISR_HIP_TIE_HALF_EVEN;
    {   // 32-bit path of k_resize: upscale (S <= 2 D keeps n + den >= 0), 16-bit divisors, every product below 2^31
        const long long denx = 2LL * R.Dx, deny = 2LL * R.Dy, den = denx * deny;
        const bool ok = R.Sx <= 2 * R.Dx && R.Sy <= 2 * R.Dy && denx < 65536 && deny < 65536 && 2 * den < 65536 && sw < (1 << 20) && sh < (1 << 20) &&
                        2 * den * 65535 + den < (1LL << 31) && (2LL * dw + 1) * R.Sx + denx < (1LL << 31) && (2LL * dh + 1) * R.Sy + deny < (1LL << 31);
        R.narrow = ok ? 1 : 0;
        R.rdenx = 1.0f / (float)denx; R.rdeny = 1.0f / (float)deny; R.rden2 = 1.0f / (float)(2 * den);
    }
    return R;
}

template <typename TIn, typename TOut>
void launch_resize(raisr_hip_ctx* c, hipStream_t s, const void* src, void* dst, const ResizeParams& R, const char* name, unsigned nz = 1)
{
    int slot;
    timer_begin(c, name, s, slot);
    if (R.sstep != 1 || R.dstep != 1) {                      // one channel of an interleaved plane: the generic kernel addresses by element step
        dim3 grid((R.dw + 63) / 64, (R.dh + 3) / 4, nz);
        hipLaunchKernelGGL((k_resize<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else if (R.dw == 2 * R.sw && R.dh == 2 * R.sh) {
        dim3 grid(((R.dw + 7) / 8 + 63) / 64, (R.sh + 3) / 4, nz);       // a thread: 8 x 2 output pixels
        hipLaunchKernelGGL((k_resize2x<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else if (2 * R.dw == 3 * R.sw && 2 * R.dh == 3 * R.sh) {
        dim3 grid(((R.dw + 11) / 12 + 63) / 64, ((R.dh + 2) / 3 + 3) / 4, nz);   // a thread: 12 x 3 output pixels
        hipLaunchKernelGGL((k_resize3x2<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else if (R.dw == R.sw && R.dh == R.sh) {
        dim3 grid((R.dw + 63) / 64, (R.dh + 3) / 4, nz);
        hipLaunchKernelGGL((k_copy<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else {
        dim3 grid((R.dw + 63) / 64, (R.dh + 3) / 4, nz);
        hipLaunchKernelGGL((k_resize<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    }
    timer_end(c, s, slot);
}

PassParams make_pass(raisr_hip_ctx* c, int pass, int W, int H)
{
    PassParams P{};
    const raisr_hip_config& g = c->cfg;
    const ModelDev& m = c->model[pass];
    P.W = W; P.H = H;
    P.lr_pitch = W; P.hash_pitch = W; P.hr_pitch = W;
    P.lo = (float)g.clamp_lo; P.hi = (float)g.clamp_hi;
    P.ilo = g.clamp_lo; P.ihi = g.clamp_hi;
    column_plan(W, g.hash_variant, P);
    P.hash2 = c->d_hash2[pass];
    P.pixel_types = m.h.pixel_types;
    P.randomness = c->blending == RAISR_HIP_BLEND_RANDOMNESS;
    P.qangle = m.h.qangle;
    P.qs0 = m.h.qstr[0]; P.qs1 = m.h.qstr[1];
    P.qc0 = m.h.qcoh[0]; P.qc1 = m.h.qcoh[1];
    P.bank = (const float*)((const char*)m.blob + kBlobHeader);
    P.bank_bytes = (int)blob_f32_bytes(m.h.hashkeys * m.h.pixel_types);
    P.bank_lm = m.bank_lm;
    P.tab14 = c->d_tab14;
    P.c1tab = c->d_c1tab;
    {   // class-1 rule (kernels_hash_certify.h): a window with L2 >= 0 has coherence >= 0.985 (AVX2 flavour: 0.964) -- index 2 needs both
        // thresholds below that (every shipped model: <= 0.48)
        const float qmax = m.h.qcoh[0] > m.h.qcoh[1] ? m.h.qcoh[0] : m.h.qcoh[1];
        P.c1_ok = (qmax < 0.98f ? 1 : 0) | (qmax < 0.96f ? 2 : 0);
    }
    P.lut_legacy = c->d_lut;
    P.zero_bucket[0] = m.zero_bucket[0]; P.zero_bucket[1] = m.zero_bucket[1];
    P.gauss_dev = c->d_gauss;
    P.asym = (m.asym_rows > 0) ? m.d_asym : nullptr;
    return P;
}

// census blend of a range of 16-row tile rows (first: P.tile_y0): the four-columns-per-lane kernel when the planes' rows are 4-sample
// aligned (every video size), the 64 x 16 LDS-tile kernel otherwise
template <typename TOut>
void launch_blend(raisr_hip_ctx* c, hipStream_t s, const void* lrp, const float* hr, const PassParams& P, void* out, int out_pitch_elems, unsigned tile_rows, unsigned nz)
{
    int slot;
    timer_begin(c, "k_blend", s, slot);
    const int rw = blend4_fits<TOut>(lrp, hr, out, out_pitch_elems, P, sizeof(float)) ? c->blend_rows : 0;
    const unsigned W = (unsigned)P.W;
    if (rw == 4) hipLaunchKernelGGL((k_blend4<TOut, 4>), dim3((W + 255) / 256, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, (TOut*)out, out_pitch_elems);
    else if (rw == 8) hipLaunchKernelGGL((k_blend4<TOut, 8>), dim3((W + 511) / 512, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, (TOut*)out, out_pitch_elems);
    else if (rw == 16) hipLaunchKernelGGL((k_blend4<TOut, 16>), dim3((W + 1023) / 1024, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, (TOut*)out, out_pitch_elems);
    else hipLaunchKernelGGL((k_blend<TOut>), dim3((W + 63) / 64, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, (TOut*)out, out_pitch_elems);
    timer_end(c, s, slot);
}
template <typename TOut>
void launch_blend16(raisr_hip_ctx* c, hipStream_t s, const void* lrp, const uint16_t* hr, const PassParams& P, const Pass16& Q, void* out, int out_pitch_elems, unsigned tile_rows, unsigned nz)
{
    int slot;
    timer_begin(c, "k_blend16", s, slot);
    const int rw = blend4_fits<TOut>(lrp, hr, out, out_pitch_elems, P, sizeof(uint16_t)) ? c->blend_rows : 0;
    const unsigned W = (unsigned)P.W;
    if (rw == 4) hipLaunchKernelGGL((k_blend4_16<TOut, 4>), dim3((W + 255) / 256, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, Q.c_avx, (TOut*)out, out_pitch_elems);
    else if (rw == 8) hipLaunchKernelGGL((k_blend4_16<TOut, 8>), dim3((W + 511) / 512, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, Q.c_avx, (TOut*)out, out_pitch_elems);
    else if (rw == 16) hipLaunchKernelGGL((k_blend4_16<TOut, 16>), dim3((W + 1023) / 1024, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, Q.c_avx, (TOut*)out, out_pitch_elems);
    else hipLaunchKernelGGL((k_blend16<TOut>), dim3((W + 63) / 64, tile_rows, nz), dim3(256), 0, s, (const TOut*)lrp, hr, P, Q, (TOut*)out, out_pitch_elems);
    timer_end(c, s, slot);
}

// one RAISR pass on an LR plane already resident in c->d_lr[pass]
// Row chunks of the LAST pass for the host-plane entry (RowsDone): the fused certified kernel and the census blend are launched on
// `nchunks` ranges of tile rows, and after each range `done(first_row, row_count)` runs (it enqueues the download of those
// output rows), so that the copy engine returns finished rows while the next rows are still being computed.  Blend tile row b
// (pixel rows [16 b, 16 b + 16)) reads HR rows up to 16 b + 16, i.e. filter tile rows <= b: the same ranges serve both kernels.
struct NoRowsDone { void operator()(int, int) const {} };

#ifdef RAISR_HIP_DEV
// development builds: profiling variants of the fused kernel's launch (parts of the kernel, rejected experiments: docs/EXPERIMENTS.md);
// returns true when one of them took the launch
template <typename TOut>
bool launch_dev_variant(raisr_hip_ctx* c, hipStream_t s, int pass, const void* lrp, PassParams& P, dim3 gf, int H, bool sym)
{
    static const int part = getenv("RAISR_HIP_AC_PART") ? atoi(getenv("RAISR_HIP_AC_PART")) : 0;
    if (part == 2 && getenv("RAISR_HIP_AC_PATTERN")) P.cert_check = atoi(getenv("RAISR_HIP_AC_PATTERN"));
    if (part == 1) { hipLaunchKernelGGL((k_hashfilter_ac<TOut, 1>), gf, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], FixAc{}); return true; }
    if (part == 2) {
        if (sym) hipLaunchKernelGGL((k_hashfilter_ac<TOut, 2, 4, true>), gf, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
        else hipLaunchKernelGGL((k_hashfilter_ac<TOut, 2>), gf, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
        return true;
    }
#ifdef RAISR_EXP_PERSIST
    if (getenv("RAISR_HIP_PERSIST") || getenv("RAISR_HIP_PERSIST_DYN")) {
        const bool dyn = !getenv("RAISR_HIP_PERSIST");
        static unsigned* ctr = nullptr;         // experiment only: one set of counters, one lane
        if (dyn) { if (!ctr) (void)hipMalloc((void**)&ctr, 16 * sizeof(unsigned)); (void)hipMemsetAsync(ctr, 0, 16 * sizeof(unsigned), s); }
        const unsigned nt = gf.x * gf.y, per = (unsigned)(c->n_cus * atoi(getenv(dyn ? "RAISR_HIP_PERSIST_DYN" : "RAISR_HIP_PERSIST")));
        hipLaunchKernelGGL((k_hashfilter_acp<TOut>), dim3(nt < per ? nt : per), dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], (int)gf.x, (int)gf.y,
                           c->n_cus, (!dyn && getenv("RAISR_HIP_PERSIST_SKEW")) ? atoi(getenv("RAISR_HIP_PERSIST_SKEW")) : 0, dyn ? ctr : (unsigned*)nullptr);
        return true;
    }
#endif
#ifdef RAISR_EXP_TILE8
    if (getenv("RAISR_HIP_TILE8")) {
        hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0, 2>), dim3(gf.x, (unsigned)((H - 2 * kMargin + 7) / 8)), dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
        return true;
    }
#endif
    return false;
}
#endif

// The production kernel of the fp32 numerics on a grid of tiles (whole plane x frames of a batch, or a range of tile rows starting
// at P.tile_y0), followed -- deferred exact path -- by k_fix_ac on the same tiles.  The self-check mode (cert_check: every pixel also
// takes the exact path) and RAISR_HIP_DEFER=0 use the in-tile worklist variant.
template <typename TOut>
void launch_hashfilter_ac(raisr_hip_ctx* c, hipStream_t s, int pass, const void* lrp, const PassParams& P, dim3 grid, bool sym, dim3 plane_tiles)
{
    int slot;
#ifdef RAISR_HIP_TESTHOOKS
    if (c->defer && !P.cert_check) {
        FixAc F = c->fixac;
        F.tiles_x = (int)plane_tiles.x;
        F.zs_tiles = plane_tiles.x * plane_tiles.y;
        timer_begin(c, "k_hashfilter_ac", s, slot);
        if (sym) hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0, 4, true, true>), grid, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], F);
        else hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0, 4, false, true>), grid, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], F);
        timer_end(c, s, slot);
        const unsigned tile_first = (unsigned)P.tile_y0 * plane_tiles.x, tile_count = grid.x * grid.y;
        timer_begin(c, "k_fix_ac", s, slot);
        hipLaunchKernelGGL((k_fix_ac<TOut>), dim3((tile_count + 4u * kFixTiles - 1u) / (4u * kFixTiles), 1, grid.z), dim3(256), 0, s, (const TOut*)lrp, P, F, c->d_hash[pass], c->d_hr[pass], tile_first, tile_count);
        timer_end(c, s, slot);
        return;
    }
#endif
    timer_begin(c, "k_hashfilter_ac", s, slot);
    if (sym) hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0, 4, true>), grid, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], FixAc{});
    else hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0>), grid, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass], FixAc{});
    timer_end(c, s, slot);
}

// frame batches: plane strides of this pass's launches (0 for a single frame); zs_out = stride of `out`'s planes
void batch_strides(const raisr_hip_ctx* c, int pass, size_t zs_out, PassParams& P)
{
    if (c->zb_n <= 1) return;
    const size_t plane = (size_t)c->passW[pass] * c->passH[pass];
    P.zs_lr = (pass == 0 && c->lr0_alias) ? c->zb_in_stride : plane;
    P.zs_hr = plane; P.zs_hash = plane; P.zs_out = zs_out;
}

template <typename TOut, typename RowsDone = NoRowsDone>
void run_pass(raisr_hip_ctx* c, hipStream_t s, int pass, void* out, int out_pitch_elems, int nchunks = 1, RowsDone done = RowsDone(), size_t zs_out = 0)
{
    const void* lrp = (pass == 0 && c->lr0_alias) ? c->lr0_alias : c->d_lr[pass];    // two-pass mode 2: pass 1 reads the caller's plane in place
    const int W = c->passW[pass], H = c->passH[pass];
    PassParams P = make_pass(c, pass, W, H);
    batch_strides(c, pass, zs_out, P);
    P.out_shift = out == c->final_out ? c->sample_shift : 0;
    const unsigned nz = (unsigned)c->zb_n;
    int slot;
    if (P.c_final > kMargin && H > 2 * kMargin) {
        constexpr int R = 4;        // rows per lane: 3..6 measure the same within noise, 8 is slower (occupancy)
        dim3 gh((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 4 * R - 1) / (4 * R));
        dim3 gf((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 15) / 16);
        const bool avx2all = !(P.a_end > P.a_begin);        // asm=avx2: no 16-wide chunks at all
        const int asym_rows = c->model[pass].asym_rows;
        bool sym = c->sym && asym_rows >= 0 && asym_rows <= c->sym_max_rows;   // symmetric filter stage of k_hashfilter_ac
#ifdef RAISR_HIP_DEV
        // TIMING PROBE, output wrong for pixels on non-palindromic rows: the symmetric stage whatever the bank, nothing redone -- what
        // a free treatment of those rows would be worth (docs/EXPERIMENTS.md)
        if (getenv("RAISR_HIP_SYM_IGNORE_ASYM")) { sym = true; P.asym = nullptr; }
#endif
#ifdef RAISR_HIP_TESTHOOKS
        if (c->fast || (c->fused && c->certify && c->split)) {
            P.cert_stats = c->d_cert_stats;
            P.cert_check = c->cert_check;
            FixLists F = c->fix;
            if (c->cert_check) {                     // self-check: remember which buckets were certified
                if (!c->fix.cert_mask && hipMalloc((void**)&c->fix.cert_mask, (size_t)c->cfg.out_width * c->cfg.out_height) != hipSuccess) c->fix.cert_mask = nullptr;
                F.cert_mask = c->fix.cert_mask;
            } else F.cert_mask = nullptr;
            F.tiles_x = (int)gf.x; F.tiles_y = (int)gf.y;
            const unsigned ntiles = gf.x * gf.y;
            const unsigned npers = ntiles < 1024u ? ntiles : 1024u;          // 4 workgroups per CU resident (LDS), each walks ~ntiles/1024 tiles
            timer_begin(c, "k_hash_ac", s, slot);
            hipLaunchKernelGGL((k_hash_ac<TOut>), dim3(npers), dim3(256), 0, s, (const TOut*)lrp, P, c->sep, F, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            if (c->fast < 2) {
            timer_begin(c, "k_fix", s, slot);
            hipLaunchKernelGGL((k_fix_sparse<TOut>), dim3((ntiles + 3u) / 4u), dim3(256), 0, s, (const TOut*)lrp, P, F, c->d_hash[pass], c->d_hash2[pass]);
            const unsigned nd = (unsigned)(gf.x * gf.y < 2048u ? gf.x * gf.y : 2048u);
            if (!avx2all)
                hipLaunchKernelGGL((k_fix_dense<TOut, false>), dim3(nd), dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, F, c->d_hash[pass], c->d_hash2[pass]);
            else
                hipLaunchKernelGGL((k_fix_dense<TOut, true>), dim3(nd), dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, F, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            }
#ifdef RAISR_HIP_DEV
            if (c->fast && c->model[pass].bank_mfma) {             // the panels are allocated by configure / set_fast (errors reported there)
                ModelDev& m = c->model[pass];
                if (!m.bank_mfma_valid) {
                    hipLaunchKernelGGL(k_build_mfma_bank, dim3((unsigned)((kMfBankHalfs + 255) / 256)), dim3(256), 0, s, P.bank, m.bank_mfma);
                    m.bank_mfma_valid = true;
                }
                {
                    dim3 gm((P.c_final - kMargin + kMfW - 1) / kMfW, (H - 2 * kMargin + kMfH - 1) / kMfH);
                    timer_begin(c, "k_filter_mfma", s, slot);
#ifdef RAISR_HIP_DEV                                                  /* profiling aid of a development build (scripts/build_exp.sh dev -DRAISR_HIP_DEV) */
                    static const int mpart = getenv("RAISR_HIP_MF_PART") ? atoi(getenv("RAISR_HIP_MF_PART")) : 0;
                    if (mpart == 1) hipLaunchKernelGGL((k_filter_mfma<TOut, 1>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    else if (mpart == 2) hipLaunchKernelGGL((k_filter_mfma<TOut, 2>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    else
#endif
                    hipLaunchKernelGGL((k_filter_mfma<TOut>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    timer_end(c, s, slot);
                }
#else
            if (0) {
#endif
            } else if (c->lds_filter && c->cfg.bits <= 10) {      // samples above 10 bits are not exact in binary16: k_filter
                const bool sp2 = P.pixel_types == 4;
                const int wh = sp2 ? 41 : 26;
                const size_t sh16 = (size_t)217 * 128 * 4 + 2 * (size_t)wh * (sp2 ? 286 : 154) * 2 + 4 * 1024;
                timer_begin(c, "k_filter_lds16", s, slot);
                if (sp2) hipLaunchKernelGGL((k_filter_lds16<TOut, 2>), dim3(c->n_cus), dim3(1024), sh16, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                else hipLaunchKernelGGL((k_filter_lds16<TOut, 1>), dim3(c->n_cus), dim3(1024), sh16, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                timer_end(c, s, slot);
            } else {
                timer_begin(c, "k_filter", s, slot);
                hipLaunchKernelGGL((k_filter<TOut>), gf, dim3(256), 0, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                timer_end(c, s, slot);
            }
        } else
#endif
        if (c->fused && c->certify && nchunks > 1 && !P.randomness && (int)gf.y >= 2 * nchunks) {
            P.write_hash = c->keep_hash_plane;
            P.cert_stats = c->d_cert_stats;
            P.cert_check = c->cert_check;
            const int Ty = (int)gf.y, Tb = (H + 15) / 16;
            for (int i = 0; i < nchunks; i++) {
                const int t0 = (int)((long long)Ty * i / nchunks), t1 = (int)((long long)Ty * (i + 1) / nchunks);
                const int b1 = i == nchunks - 1 ? Tb : t1;
                P.tile_y0 = t0;
                launch_hashfilter_ac<TOut>(c, s, pass, lrp, P, dim3(gf.x, (unsigned)(t1 - t0)), sym, gf);
                launch_blend<TOut>(c, s, lrp, (const float*)c->d_hr[pass], P, out, out_pitch_elems, (unsigned)(b1 - t0), 1u);
                const int r0 = 16 * t0, r1 = i == nchunks - 1 ? H : 16 * t1;
                done(r0, r1 - r0);
            }
            return;
        } else if (c->fused && c->certify) {
            P.write_hash = c->keep_hash_plane;
            P.cert_stats = c->d_cert_stats;
            P.cert_check = c->cert_check;
#ifdef RAISR_HIP_DEV
            timer_begin(c, "k_hashfilter_ac", s, slot);
            const bool dev_taken = launch_dev_variant<TOut>(c, s, pass, lrp, P, gf, H, sym);
            timer_end(c, s, slot);
            if (!dev_taken)
#endif
            launch_hashfilter_ac<TOut>(c, s, pass, lrp, P, dim3(gf.x, gf.y, nz), sym, gf);
        } else if (c->fused) {
            P.write_hash = c->keep_hash_plane;
            timer_begin(c, "k_hashfilter", s, slot);
            if (!avx2all)
                hipLaunchKernelGGL((k_hashfilter<TOut, false>), gf, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->d_hash[pass], c->d_hr[pass]);
            else
                hipLaunchKernelGGL((k_hashfilter<TOut, true>), gf, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->d_hash[pass], c->d_hr[pass]);
            timer_end(c, s, slot);
        } else {
            timer_begin(c, "k_hash", s, slot);
            if (!avx2all)
                hipLaunchKernelGGL((k_hash<R, TOut, false>), gh, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->d_hash[pass], c->d_hash2[pass]);
            else
                hipLaunchKernelGGL((k_hash<R, TOut, true>), gh, dim3(256), 0, s, (const TOut*)lrp, P, c->gauss, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            timer_begin(c, "k_filter", s, slot);
            hipLaunchKernelGGL((k_filter<TOut>), gf, dim3(256), 0, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass]);
            timer_end(c, s, slot);
        }
    }
    if (P.randomness) {
        dim3 gb((W + 63) / 64, (H + 3) / 4);
        timer_begin(c, "k_blend_rand", s, slot);
        hipLaunchKernelGGL((k_blend_rand<TOut, false>), gb, dim3(256), 0, s, (const TOut*)lrp, (const void*)c->d_hr[pass], P, (TOut*)out, out_pitch_elems);
        timer_end(c, s, slot);
        done(0, H);
        return;
    }
#ifdef RAISR_HIP_DEV                                                  /* ceiling of a fused blend epilogue: the frame rate with k_blend gone (output wrong) */
    static const bool skip_blend = getenv("RAISR_HIP_SKIP_BLEND") != nullptr;
    if (skip_blend) { done(0, H); return; }
#endif
    launch_blend<TOut>(c, s, lrp, (const float*)c->d_hr[pass], P, out, out_pitch_elems, (unsigned)((H + 15) / 16), nz);
    done(0, H);
}

// per-pass constants of the binary16 pipeline
Pass16 make_pass16(raisr_hip_ctx* c, int pass, int W)
{
    const ModelDev& m = c->model[pass];
    Pass16 Q{};
    const int rows = m.h.hashkeys * m.h.pixel_types;
    Q.bank16 = (const uint32_t*)((const char*)m.blob + kBlobHeader + blob_f32_bytes(rows));
    Q.bank16_bytes = (int)blob_f16_bytes(rows);
    Q.tab16 = c->d_tab16;
    Q.qangle = m.h.qangle16; Q.qs0 = m.h.qstr16[0]; Q.qs1 = m.h.qstr16[1]; Q.qc0 = m.h.qcoh16[0]; Q.qc1 = m.h.qcoh16[1];
    Q.ls0 = m.ls16[0]; Q.ls1 = m.ls16[1]; Q.cm0 = m.cm16[0]; Q.cm1 = m.cm16[1]; Q.ct0 = m.ct16[0]; Q.ct1 = m.ct16[1];
    Q.fold = (c->fold16 && m.fold16) ? 1 : 0;
    {   // NF_8 / NF_10 (Raisr_globals.h:208-209; Raisr_AVX512FP16.cpp:146-151)
        const float maxv = c->cfg.bits == 8 ? 255.0f : 1023.0f;
        volatile float nf = 1.0f / (maxv * maxv * 2.0f * 2.0f);
        Q.nf = nf;
    }
    Q.c_avx = (W - 1) - ((W - 1) % 32) + 1;
    return Q;
}

// one pass of the AVX512-FP16-exact pipeline (binary16 arithmetic)
template <typename TOut>
void run_pass16(raisr_hip_ctx* c, hipStream_t s, int pass, void* out, int out_pitch_elems, size_t zs_out = 0)
{
    const void* lrp = (pass == 0 && c->lr0_alias) ? c->lr0_alias : c->d_lr[pass];    // two-pass mode 2: pass 1 reads the caller's plane in place
    const int W = c->passW[pass], H = c->passH[pass];
    PassParams P = make_pass(c, pass, W, H);
    batch_strides(c, pass, zs_out, P);
    P.out_shift = out == c->final_out ? c->sample_shift : 0;
    const unsigned nz = (unsigned)c->zb_n;
    const Pass16 Q = make_pass16(c, pass, W);
    int slot;
    if (P.c_final > kMargin && H > 2 * kMargin) {
        dim3 gh((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 15) / 16, c->fused ? nz : 1u);
        if (c->fused) {
            P.write_hash = c->keep_hash_plane;
            timer_begin(c, "k_hashfilter16", s, slot);
            hipLaunchKernelGGL((k_hashfilter16<TOut>), gh, dim3(256), 0, s, (const TOut*)lrp, P, Q, c->gauss16, c->d_hash[pass], (uint16_t*)c->d_hr[pass]);
            timer_end(c, s, slot);
        } else {
            timer_begin(c, "k_hash16", s, slot);
            hipLaunchKernelGGL((k_hash16<4, TOut>), gh, dim3(256), 0, s, (const TOut*)lrp, P, Q, c->gauss16, c->d_hash[pass]);
            timer_end(c, s, slot);
            timer_begin(c, "k_filter16", s, slot);
            hipLaunchKernelGGL((k_filter16<TOut>), gh, dim3(256), 0, s, (const TOut*)lrp, (const uint8_t*)c->d_hash[pass], P, Q, (uint16_t*)c->d_hr[pass]);
            timer_end(c, s, slot);
        }
    }
    if (P.randomness) {
        dim3 gr((W + 63) / 64, (H + 3) / 4);
        timer_begin(c, "k_blend_rand", s, slot);
        hipLaunchKernelGGL((k_blend_rand<TOut, true>), gr, dim3(256), 0, s, (const TOut*)lrp, (const void*)c->d_hr[pass], P, (TOut*)out, out_pitch_elems);
        timer_end(c, s, slot);
        return;
    }
    launch_blend16<TOut>(c, s, lrp, (const uint16_t*)c->d_hr[pass], P, Q, out, out_pitch_elems, (unsigned)((H + 15) / 16), nz);
}

void free_scratch(raisr_hip_ctx* c)
{
    for (int i = 0; i < 2; i++) {
        if (c->d_lr[i]) (void)hipFree(c->d_lr[i]);
        if (c->d_hash[i]) (void)hipFree(c->d_hash[i]);
        if (c->d_hash2[i]) (void)hipFree(c->d_hash2[i]);
        if (c->d_hr[i]) (void)hipFree(c->d_hr[i]);
        c->d_lr[i] = nullptr; c->d_hash[i] = nullptr; c->d_hash2[i] = nullptr; c->d_hr[i] = nullptr;
    }
    if (c->d_mid) (void)hipFree(c->d_mid);
    c->d_mid = nullptr;
#ifdef RAISR_HIP_TESTHOOKS
    if (c->fix.counters) (void)hipFree(c->fix.counters);
    if (c->fix.counts) (void)hipFree(c->fix.counts);
    if (c->fix.sparse) (void)hipFree(c->fix.sparse);
    if (c->fix.dense) (void)hipFree(c->fix.dense);
    if (c->fix.cert_mask) (void)hipFree(c->fix.cert_mask);
    if (c->fixac.counts) (void)hipFree(c->fixac.counts);
    if (c->fixac.entries) (void)hipFree(c->fixac.entries);
    c->fixac = FixAc{};
    c->fix = FixLists{};
    c->fix_tiles = 0;
#endif
}

}  // namespace

extern "C" {

const char* raisr_hip_last_error(void) { return g_err.c_str(); }
void raisr_hip_destroy(raisr_hip_ctx* c);
const char* raisr_hip_version(void) { return "raisr-hip 0.1 (gfx950)"; }

int raisr_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Streams and the host-plane staging buffer are recycled through process-wide pools instead of being destroyed with
// their context: hosts such as FFmpeg re-create the filter per clip, and every create/destroy cycle should leave the
// device exactly as it found it (tests/test_gpu_host_api.py::test_context_lifecycle_does_not_leak_device_memory).
static std::mutex g_pool_mu;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;      // (device, idle stream)

static int pool_get_stream(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_stream_pool.size(); i++)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return RAISR_HIP_OK;
            }
    }
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return RAISR_HIP_OK;
}

static void pool_put_stream(int device, hipStream_t s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_stream_pool.emplace_back(device, s);
}

struct StageBuf { int device; void* ptr; size_t bytes; };
static std::vector<StageBuf> g_stage_pool;

static void* pool_get_stage(int device, size_t need, size_t* got)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_stage_pool.size(); i++)
            if (g_stage_pool[i].device == device && g_stage_pool[i].bytes >= need) {
                void* p = g_stage_pool[i].ptr; *got = g_stage_pool[i].bytes;
                g_stage_pool.erase(g_stage_pool.begin() + (long)i);
                return p;
            }
    }
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) return nullptr;
    *got = need;
    return p;
}

static void pool_put_stage(int device, void* p, size_t bytes)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_stage_pool.size() >= 16) {               // bound the idle set: drop the smallest buffer
        size_t k = 0;
        for (size_t i = 1; i < g_stage_pool.size(); i++) if (g_stage_pool[i].bytes < g_stage_pool[k].bytes) k = i;
        if (g_stage_pool[k].bytes < bytes) { (void)hipFree(g_stage_pool[k].ptr); g_stage_pool[k] = {device, p, bytes}; }
        else (void)hipFree(p);
        return;
    }
    g_stage_pool.push_back({device, p, bytes});
}

static int create_impl(raisr_hip_ctx* c)
{
    if (const char* e = getenv("RAISR_HIP_FUSED")) c->fused = atoi(e) != 0;       // A/B switch: 0 = separate k_hash + k_filter
    if (const char* e = getenv("RAISR_HIP_CERTIFY")) c->certify = atoi(e) != 0;   // A/B switch: 0 = exact tensor for every pixel
#ifdef RAISR_HIP_TESTHOOKS
    if (const char* e = getenv("RAISR_HIP_DEFER")) c->defer = atoi(e) != 0;       // comparison pipeline: 1 = k_fix_ac instead of the in-tile worklist
#else
    for (const char* v : {"RAISR_HIP_DEFER", "RAISR_HIP_SPLIT"})                  // asked for, not in the product library: say so
        if (const char* e = getenv(v)) if (atoi(e) > 0) return fail(RAISR_HIP_EINVAL, "RAISR_HIP_DEFER / RAISR_HIP_SPLIT select comparison pipelines that exist in builds with -DRAISR_HIP_TESTHOOKS only (libraisr_hip_testhooks.so)");
#endif
    if (const char* e = getenv("RAISR_HIP_FOLD16")) c->fold16 = atoi(e) != 0;
    if (const char* e = getenv("RAISR_HIP_SYM")) c->sym = atoi(e) != 0;           // A/B switch: 0 = eight coefficient loads per pixel whatever the bank
    if (const char* e = getenv("RAISR_HIP_SYM_MAX_ROWS")) c->sym_max_rows = atoi(e);
    if (const char* e = getenv("RAISR_HIP_BLEND_ROWS")) { const int v = atoi(e); c->blend_rows = (v == 4 || v == 8 || v == 16) ? v : 0; }   // A/B switch: 0 = the LDS-tile blend kernels
#ifdef RAISR_HIP_DEV
    if (const char* e = getenv("RAISR_HIP_FAST")) { const int v = atoi(e); c->fast = v < 0 ? 0 : (v > 2 ? 2 : v); }         // NON-bit-exact fast mode (see raisr_hip_set_fast)
#else
    if (const char* e = getenv("RAISR_HIP_FAST")) if (atoi(e) > 0) return fail(RAISR_HIP_EINVAL, kNoFastMode);           // asked for, not available: say so
#endif
#ifdef RAISR_HIP_TESTHOOKS
    if (const char* e = getenv("RAISR_HIP_SPLIT")) c->split = atoi(e) != 0;       // comparison pipeline: 1 = k_hash_ac + filter kernel
#endif
    if (const char* e = getenv("RAISR_HIP_LDS_FILTER")) c->lds_filter = atoi(e) != 0;
    if (const char* e = getenv("RAISR_HIP_CHUNKS")) { const int v = atoi(e); c->chunks = v < 1 ? 1 : (v > 8 ? 8 : v); }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount >= 4) c->n_cus = prop.multiProcessorCount & ~3;
        // k_filter_lds16 declares ~160 KB of dynamic LDS
#ifdef RAISR_HIP_DEV
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
#endif
#ifdef RAISR_HIP_DEV
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
#endif
#ifdef RAISR_HIP_TESTHOOKS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint8_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint8_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint16_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
#endif
    }
    HIP_TRY(hipMalloc((void**)&c->d_gauss, sizeof(GaussW)));
    int rc = pool_get_stream(c->device, &c->stream);
    if (rc) return rc;
    rc = pool_get_stream(c->device, &c->stream2);
    if (rc) return rc;
    // small shared tables
    std::vector<uint2> tab(128);
    for (int i = 0; i < 64; i++) { tab[i] = make_uint2(X86_RCP14_C0[i], X86_RCP14_C1[i]); tab[64 + i] = make_uint2(X86_RSQRT14_C0[i], X86_RSQRT14_C1[i]); }
    std::vector<uint16_t> lut(4096);
    for (int i = 0; i < 2048; i++) { lut[i] = X86_RCP_LUT[i]; lut[2048 + i] = X86_RSQRT_LUT[i]; }
    HIP_TRY(hipMalloc((void**)&c->d_tab14, tab.size() * sizeof(uint2)));
    HIP_TRY(hipMalloc((void**)&c->d_lut, lut.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(c->d_tab14, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_lut, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    {   // class-1 sign table of the certified hash stage, from the exact device models (RAISR_HIP_C1=0: the class goes to the worklist as before round 6)
        const char* e = getenv("RAISR_HIP_C1");
        if (!(e && e[0] == '0')) {
            HIP_TRY(hipMalloc((void**)&c->d_c1tab, kC1Buckets));
            hipLaunchKernelGGL(k_build_c1tab, dim3(kC1Buckets / 256), dim3(256), 0, c->stream, c->d_tab14, c->d_c1tab);
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    std::vector<uint16_t> t16(3072 + 2048);
    for (int i = 0; i < 1024; i++) { t16[i] = X86_RCPPH_T[i]; t16[1024 + i] = X86_RSQRTPH_T0[i]; t16[2048 + i] = X86_RSQRTPH_T1[i]; }
    // composite VRCPPH(VRSQRTPH(x)) for positive finite x (kernels_fp16.h, sqrt_ph): VRSQRTPH gives mantissa + exponent te of row t
    // (less the halved input exponent), VRCPPH of that the row t2 of its mantissa with exponent te2 + 15 - (te - half); folded:
    // C[p][m] = ((te2 + 15 - te) << 10) | mantissa(t2), result = C + (half << 10).  te2 + 15 - te is 13..16 for every row.
    for (int p = 0; p < 2; p++)
        for (int m = 0; m < 1024; m++) {
            const unsigned t = t16[1024 + 1024 * p + m], t2 = t16[t & 1023u];
            const int base = (int)((t2 >> 10) & 31u) + 15 - (int)((t >> 10) & 31u);
            if (base < 8 || base > 23) return fail(RAISR_HIP_ERUNTIME, "composite square-root table: exponent out of the expected range");
            t16[3072 + 1024 * p + m] = (uint16_t)(((unsigned)base << 10) | (t2 & 1023u));
        }
    HIP_TRY(hipMalloc((void**)&c->d_tab16, t16.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(c->d_tab16, t16.data(), t16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());           // (null-stream copies are not ordered with the non-blocking streams the kernels use)
    {   // un-normalised Gaussian in binary16, (fp16)literal (Raisr_globals.h:267-278)
        for (int i = 0; i < 11; i++)
            for (int k = 0; k < 11; k++) {
                const _Float16 wv = (_Float16)kGaussQ[i < 6 ? i : 10 - i][k < 6 ? k : 10 - k];
                uint16_t u; memcpy(&u, &wv, 2);
                c->gauss16.wT[k][i] = (uint32_t)u | ((uint32_t)u << 16);
            }
    }
    return RAISR_HIP_OK;
}


int raisr_hip_create(raisr_hip_ctx** out, int device_index)
{
    if (!out) return fail(RAISR_HIP_EINVAL, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RAISR_HIP_ENODEV, "no HIP device visible", e);
    if (device_index < 0 || device_index >= n) return fail(RAISR_HIP_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device_index));
    raisr_hip_ctx* c = new raisr_hip_ctx();
    c->device = device_index;
    {   // NUMA node of the device (several-node hosts only): its row-copy pool runs on that node's CPUs (host_copy.h, raisr_numa)
        char bdf[32] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device_index) == hipSuccess) c->numa_node = raisr_numa::node_for_device(bdf);
        else (void)hipGetLastError();
        c->bounce.node = c->numa_node;
    }
    const int rc = create_impl(c);
    if (rc != RAISR_HIP_OK) { const std::string keep = g_err; raisr_hip_destroy(c); g_err = keep; return rc; }
    *out = c;
    return RAISR_HIP_OK;
}

void raisr_hip_destroy(raisr_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->timer.pool) if (e) (void)hipEventDestroy(e);
    free_scratch(c);
    for (int i = 0; i < 2; i++) if (c->model[i].blob) (void)hipFree(c->model[i].blob);
    for (int i = 0; i < 2; i++) if (c->model[i].bank_mfma) (void)hipFree(c->model[i].bank_mfma);
    for (int i = 0; i < 2; i++) if (c->model[i].d_asym) (void)hipFree(c->model[i].d_asym);
    for (int i = 0; i < 2; i++) if (c->model[i].bank_lm) (void)hipFree(c->model[i].bank_lm);
    if (c->d_tab14) (void)hipFree(c->d_tab14);
    if (c->d_c1tab) (void)hipFree(c->d_c1tab);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_tab16) (void)hipFree(c->d_tab16);
    if (c->d_cert_stats) (void)hipFree(c->d_cert_stats);
    if (c->ev_chroma) (void)hipEventDestroy(c->ev_chroma);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->ev_comp) (void)hipEventDestroy(c->ev_comp);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->ev_kern) (void)hipEventDestroy(c->ev_kern);
    for (hipEvent_t& e : c->ev_chunk) if (e) (void)hipEventDestroy(e);
    c->bounce.release();
    if (c->own_stream) { c->stream = c->own_stream; c->own_stream = nullptr; }
    if (c->d_gauss) (void)hipFree(c->d_gauss);
    pool_put_stage(c->device, c->d_stage, c->d_stage_bytes);
    pool_put_stream(c->device, c->stream);
    pool_put_stream(c->device, c->stream2);
    delete c;
}

size_t raisr_hip_model_blob_bytes(int hashkeys, int pixel_types)
{
    if (hashkeys <= 0 || pixel_types <= 0) return 0;
    const int rows = hashkeys * pixel_types;
    return (size_t)kBlobHeader + blob_f32_bytes(rows) + blob_f16_bytes(rows);
}

int raisr_hip_pack_model_blob(void* host_blob, const float* bank, int hashkeys, int pixel_types,
                              const double qstr[2], const double qcoh[2], int quant_angle)
{
    if (!host_blob || !bank || !qstr || !qcoh) return fail(RAISR_HIP_EINVAL, "null argument");
    if (hashkeys <= 0 || hashkeys > 255 || (pixel_types != 1 && pixel_types != 4) || quant_angle <= 0)
        return fail(RAISR_HIP_EINVAL, "unsupported model geometry");
    auto hbits = [](_Float16 v) { uint16_t u; memcpy(&u, &v, 2); return u; };
    BlobHeader h{};
    h.magic = kBlobMagic; h.hashkeys = hashkeys; h.pixel_types = pixel_types; h.quant_angle = quant_angle;
    h.qangle = (float)quant_angle / 3.141592653f;           // gQAngle, Raisr.cpp:1553
    for (int i = 0; i < 2; i++) {
        h.qstr[i] = (float)qstr[i]; h.qcoh[i] = (float)qcoh[i];                 // (float)stod(token), Raisr.cpp:377,413
        h.qstr16[i] = hbits((_Float16)qstr[i]); h.qcoh16[i] = hbits((_Float16)qcoh[i]);   // (_Float16)stod(token)
    }
    h.qangle16 = hbits((_Float16)h.qangle);                 // _mm512_set1_ph(gQAngle)
    memcpy(host_blob, &h, sizeof h);
    const size_t rows = (size_t)hashkeys * pixel_types;
    float* dst = (float*)((char*)host_blob + kBlobHeader);
    memset(dst, 0, blob_f32_bytes((int)rows));
    for (size_t r = 0; r < rows; r++) memcpy(dst + r * kTapsPad, bank + r * kTaps, kTaps * sizeof(float));
    // binary16 bank (Raisr.cpp:344-350: currentfilter[j] = (_Float16)weight), lane-pair interleaved
    uint16_t* d16 = (uint16_t*)((char*)host_blob + kBlobHeader + blob_f32_bytes((int)rows));
    for (size_t r = 0; r < rows; r++)
        for (int ch = 0; ch < 4; ch++)
            for (int l = 0; l < 16; l++) {
                const int k0 = 32 * ch + l, k1 = k0 + 16;
                d16[(r * 64 + ch * 16 + l) * 2 + 0] = k0 < kTaps ? hbits((_Float16)bank[r * kTaps + k0]) : (uint16_t)0;
                d16[(r * 64 + ch * 16 + l) * 2 + 1] = k1 < kTaps ? hbits((_Float16)bank[r * kTaps + k1]) : (uint16_t)0;
            }
    return RAISR_HIP_OK;
}

// Folded thresholds of the binary16 hash (kernels_fp16.h, Pass16).  Strength: ls = the smallest binary16 x with
// qs <= fl16(x / 100) -- fl16(x / 100) is monotone in x, found by scanning all 63 490 non-NaN values with the correctly rounded
// division (fp32 quotient rounded once more to binary16: innocuous, 24 >= 2 * 11 + 2).  Coherence: cm = midpoint between qc and
// its predecessor, ct = 1 when that midpoint rounds to qc (qc's last mantissa bit is even); needs 0 < qc < inf.
static void fold16_thresholds(ModelDev& m)
{
    auto h_of = [](uint16_t u) { _Float16 v; memcpy(&v, &u, 2); return v; };
    m.fold16 = 1;
    for (int i = 0; i < 2; i++) {
        const _Float16 qs = h_of(m.h.qstr16[i]);
        bool found = false;
        _Float16 best = (_Float16)0.0f;
        for (uint32_t u = 0; u < 65536u; u++) {
            if ((u & 0x7c00u) == 0x7c00u && (u & 0x3ffu)) continue;              // NaN
            const _Float16 x = h_of((uint16_t)u);
            volatile float qf = (float)x / 100.0f;
            const _Float16 f = (_Float16)qf;
            if (qs <= f && (!found || x < best)) { best = x; found = true; }
        }
        uint16_t bits = 0x7e00;                                                   // NaN: no L1 passes
        if (found) memcpy(&bits, &best, 2);
        m.ls16[i] = bits;
        const uint16_t qc = m.h.qcoh16[i];
        if (qc == 0 || qc >= 0x7c00u) { m.fold16 = 0; continue; }               // zero, negative, infinite or NaN threshold: keep the divisions
        m.cm16[i] = 0.5f * ((float)h_of(qc) + (float)h_of((uint16_t)(qc - 1)));  // exact: both binary16, their sum has <= 12 significant bits
        m.ct16[i] = (qc & 1u) ? 0 : 1;
    }
}

// The filter stage's copy of the fp32 bank (kernels_filter.h, k_lane_major_bank), rebuilt whenever a model arrives.
static int build_lane_major_bank(raisr_hip_ctx* c, int pass_index)
{
    ModelDev& m = c->model[pass_index];
    const int rows = m.h.hashkeys * m.h.pixel_types;
    const size_t bytes = (size_t)rows * kLmRow * sizeof(float);
    if (m.bank_lm && m.bank_lm_bytes != bytes) { (void)hipFree(m.bank_lm); m.bank_lm = nullptr; }
    if (!m.bank_lm) {
        if (hipMalloc((void**)&m.bank_lm, bytes) != hipSuccess) { m.bank_lm = nullptr; return fail(RAISR_HIP_ENOMEM, "hipMalloc"); }
        m.bank_lm_bytes = bytes;
    }
    const unsigned n = (unsigned)rows * kLmRow;
    hipLaunchKernelGGL(k_lane_major_bank, dim3((n + 255u) / 256u), dim3(256), 0, c->stream,
                       (const float*)((const char*)m.blob + kBlobHeader), m.bank_lm, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RAISR_HIP_OK;
}

// Which rows of the fp32 bank are palindromes where the symmetric filter stage needs them to be (f[k] == f[120 - k] as bit patterns
// for k <= 56: the stage loads taps 0..63 directly and mirrors taps 64..120 from taps 56..0 -- the pairs (57, 63), (58, 62), (59, 61)
// are both loaded, a mismatch there costs nothing)?  Decides whether the symmetric filter stage may run for this model and lists the
// rows whose steps it runs a second time with the true coefficients (kernels_filter.h).
// `host_bank` = the blob's fp32 bank [rows][128] on the host, or null: then it is read back from the device blob.
static int scan_bank_symmetry(raisr_hip_ctx* c, int pass_index, const float* host_bank)
{
    ModelDev& m = c->model[pass_index];
    m.asym_rows = -1;
    const int rows = m.h.hashkeys * m.h.pixel_types;
    std::vector<float> back;
    if (!host_bank) {
        back.resize((size_t)rows * kTapsPad);
        HIP_TRY(hipMemcpy(back.data(), (const char*)m.blob + kBlobHeader, blob_f32_bytes(rows), hipMemcpyDeviceToHost));
        host_bank = back.data();
    }
    uint32_t bits[32] = {0};
    int n = 0;
    for (int r = 0; r < rows && r < 1024; r++) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(host_bank + (size_t)r * kTapsPad);
        bool pal = true;
        for (int k = 0; k < 57 && pal; k++) pal = f[k] == f[120 - k];
        for (int k = kTaps; k < kTapsPad && pal; k++) pal = f[k] == 0u;            // the padding must be +0 (it is: pack_model_blob)
        if (!pal) { bits[r >> 5] |= 1u << (r & 31); n++; }
    }
    if (!m.d_asym) { if (hipMalloc((void**)&m.d_asym, sizeof bits) != hipSuccess) { m.d_asym = nullptr; return fail(RAISR_HIP_ENOMEM, "hipMalloc"); } }
    HIP_TRY(hipMemcpy(m.d_asym, bits, sizeof bits, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    m.asym_rows = n;
    return RAISR_HIP_OK;
}

// Bucket of the all-zero structure tensor (flat windows) in both hash flavours, from the device's exact hash code:
// the certified hash stage looks it up instead of hashing (0, 0, 0) per pixel.
static int compute_zero_buckets(raisr_hip_ctx* c, int pass_index)
{
    ModelDev& m = c->model[pass_index];
    float* d_in = nullptr; uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, 3 * sizeof(float)));
    if (hipMalloc((void**)&d_out, 2) != hipSuccess) { (void)hipFree(d_in); return fail(RAISR_HIP_ENOMEM, "hipMalloc"); }
    int rc = RAISR_HIP_OK;
    uint8_t h[2] = {0, 0};
    if (hipMemsetAsync(d_in, 0, 3 * sizeof(float), c->stream) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemset");   // stream-ordered before k_debug_hash
    PassParams P = make_pass(c, pass_index, 0, 0);
    for (int legacy = 0; legacy < 2 && !rc; legacy++) {
        hipLaunchKernelGGL(k_debug_hash, dim3(1), dim3(256), 0, c->stream, d_in, 1u, P, legacy, d_out + legacy);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "k_debug_hash");
    }
    if (!rc && hipMemcpy(h, d_out, 2, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    (void)hipFree(d_in); (void)hipFree(d_out);
    m.zero_bucket[0] = h[0]; m.zero_bucket[1] = h[1];
    return rc;
}

int raisr_hip_set_model_blob_device(raisr_hip_ctx* c, int pass_index, const void* device_blob, size_t bytes, void* stream)
{
    if (!c || !device_blob || pass_index < 0 || pass_index > 1) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (bytes < (size_t)kBlobHeader) return fail(RAISR_HIP_EINVAL, "model blob shorter than its header");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    BlobHeader h{};
    HIP_TRY(hipMemcpyAsync(&h, device_blob, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h.magic != kBlobMagic || h.hashkeys <= 0 || h.hashkeys > 255 || (h.pixel_types != 1 && h.pixel_types != 4) ||
        raisr_hip_model_blob_bytes(h.hashkeys, h.pixel_types) != bytes)
        return fail(RAISR_HIP_EINVAL, "model blob corrupted");
    ModelDev& m = c->model[pass_index];
    if (m.blob && m.bytes != bytes) { (void)hipFree(m.blob); m.blob = nullptr; }
    if (!m.blob) { if (hipMalloc(&m.blob, bytes) != hipSuccess) return fail(RAISR_HIP_ENOMEM, "model blob alloc"); }
    HIP_TRY(hipMemcpyAsync(m.blob, device_blob, bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    m.bytes = bytes; m.h = h; m.valid = true; m.bank_mfma_valid = false;
    fold16_thresholds(m);
    if (int rc = build_lane_major_bank(c, pass_index)) return rc;
    if (int rc = scan_bank_symmetry(c, pass_index, nullptr)) return rc;
    return compute_zero_buckets(c, pass_index);
}

// Multi-GPU start-up: the one collective of the path.  RCCL is resolved on first use, so a single-GPU consumer of
// this library carries no librccl dependency; the handful of declarations below are RCCL's stable C ABI (rccl.h).
int raisr_hip_broadcast_model_blob(void* nccl_comm, int root, void* device_blob, size_t bytes, void* stream)
{
    if (!nccl_comm || !device_blob || bytes < (size_t)kBlobHeader || root < 0) return fail(RAISR_HIP_EINVAL, "bad argument");
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_fn)(int);
    static std::mutex mu;
    static bcast_fn bcast = nullptr;
    static errstr_fn errstr = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!bcast) {
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return fail(RAISR_HIP_ERUNTIME, "librccl.so not found (needed only for the multi-GPU model broadcast)");
            bcast = reinterpret_cast<bcast_fn>(dlsym(h, "ncclBroadcast"));
            errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
            if (!bcast) return fail(RAISR_HIP_ERUNTIME, "ncclBroadcast not exported by librccl");
        }
    }
    const int kNcclUint8 = 1;                                         // ncclDataType_t
    const int rc = bcast(device_blob, device_blob, bytes, kNcclUint8, root, nccl_comm, (hipStream_t)stream);
    if (rc != 0) {
        g_err = std::string("ncclBroadcast: ") + (errstr ? errstr(rc) : "error");
        return RAISR_HIP_ERUNTIME;
    }
    return RAISR_HIP_OK;
}

// In-process counterpart for ONE host process driving several GPUs (raisr_hip_stream_create_multi): blobs[0] on devices[0]
// holds the packed model; blobs[i] on devices[i] receives it (raisr_hip_broadcast_model_blob_devices below).
// In-process RCCL (opt-in, RAISR_HIP_RCCL=1): librccl is dlopen'ed on first use; ONE communicator set per device list, created on first use
// and kept for the life of the process (both passes of a model and every later model reuse it; round 5 created and destroyed one per pass).
namespace {
struct RcclApi {
    typedef int (*initall_fn)(void**, int, const int*);
    typedef int (*group_fn)(void);
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    initall_fn initall = nullptr; group_fn gstart = nullptr, gend = nullptr; bcast_fn bcast = nullptr;
    bool tried = false;
    std::map<std::vector<int>, std::vector<void*>> comms;     // device list -> communicators (never destroyed: RCCL's teardown at exit is not worth a hang)
    std::mutex mu;
};
RcclApi* rccl_api() { static RcclApi* a = new RcclApi; return a; }

// 0 = broadcast done, 1 = RCCL not available for this list (caller copies), -1 = RCCL failed mid-way
int rccl_broadcast_in_process(const int* devices, int n, void* const* blobs, size_t bytes)
{
    RcclApi& A = *rccl_api();
    std::lock_guard<std::mutex> lk(A.mu);
    if (!A.tried) {
        A.tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            A.gstart = reinterpret_cast<RcclApi::group_fn>(dlsym(h, "ncclGroupStart"));
            A.gend = reinterpret_cast<RcclApi::group_fn>(dlsym(h, "ncclGroupEnd"));
            A.bcast = reinterpret_cast<RcclApi::bcast_fn>(dlsym(h, "ncclBroadcast"));
            A.initall = reinterpret_cast<RcclApi::initall_fn>(dlsym(h, "ncclCommInitAll"));
            if (!A.gstart || !A.gend || !A.bcast) A.initall = nullptr;
        }
    }
    if (!A.initall) return 1;
    const std::vector<int> key(devices, devices + n);
    auto it = A.comms.find(key);
    if (it == A.comms.end()) {
        std::vector<void*> cm((size_t)n, nullptr);
        if (A.initall(cm.data(), n, devices) != 0) { (void)hipGetLastError(); return 1; }      // no communicator: copies
        it = A.comms.emplace(key, cm).first;
    }
    const std::vector<void*>& comms = it->second;
    const int kNcclUint8 = 1;
    bool ok = A.gstart() == 0;
    for (int i = 0; i < n && ok; i++)
        ok = hipSetDevice(devices[i]) == hipSuccess && A.bcast(blobs[i], blobs[i], bytes, kNcclUint8, 0, comms[(size_t)i], (hipStream_t) nullptr) == 0;
    ok = (A.gend() == 0) && ok;
    for (int i = 0; i < n; i++) if (hipSetDevice(devices[i]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) ok = false;
    return ok ? 0 : -1;
}
}  // namespace

// Hand the model blob that sits on devices[0] (blobs[0]) to the other devices of a one-process ring.
// DEFAULT (round 6): n - 1 concurrent peer copies out of devices[0].  xGMI on an MI355X node is a full mesh of point-to-point links, so the
// copies to different GPUs travel on different links at the same time -- for one 664 KB blob that is the optimal broadcast, a ring
// collective would serialise hops for nothing; peer access is switched on where the runtime grants it (else the runtime stages through
// the host, still correct).  The in-process RCCL broadcast (ncclCommInitAll + grouped ncclBroadcast from one thread) is OPT-IN with
// RAISR_HIP_RCCL=1: it has only ever run with one rank (no second GPU in any build or test box so far), and a hang inside it would
// stop every multi-GPU Submit at model load.  RAISR_HIP_NO_RCCL=1 still forces the copies; RAISR_HIP_FORCE_RCCL=1 (tests) sends even a
// single device through the communicator.  The one-process-per-GPU launch (bench.py --gpus N, sharding.py) broadcasts with RCCL through
// torch.distributed as before.
int raisr_hip_broadcast_model_blob_devices(const int* devices, int n, void* const* blobs, size_t bytes)
{
    if (!devices || !blobs || n < 1 || bytes < (size_t)kBlobHeader) return fail(RAISR_HIP_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) if (!blobs[i] || devices[i] < 0) return fail(RAISR_HIP_EINVAL, "bad argument");
    const char* force = getenv("RAISR_HIP_FORCE_RCCL");
    const bool forced = force && atoi(force) != 0;
    if (n == 1 && !forced) return RAISR_HIP_OK;
    bool distinct = true;
    for (int i = 0; i < n && distinct; i++)
        for (int k = 0; k < i; k++) if (devices[k] == devices[i]) { distinct = false; break; }
    const char* no = getenv("RAISR_HIP_NO_RCCL");
    const char* want = getenv("RAISR_HIP_RCCL");
    if (distinct && (forced || (want && atoi(want) != 0)) && !(no && atoi(no) != 0)) {
        const int r = rccl_broadcast_in_process(devices, n, blobs, bytes);
        if (r == 0) return RAISR_HIP_OK;
        if (r < 0) return fail(RAISR_HIP_ERUNTIME, "in-process RCCL broadcast of the model blob failed");
    }
    // peer copies out of devices[0], all in flight together (one per destination device, on that device's null stream)
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipDeviceSynchronize());                                     // blobs[0] is complete
    for (int i = 1; i < n; i++) {
        if (blobs[i] == blobs[0]) continue;
        if (devices[i] == devices[0]) { HIP_TRY(hipSetDevice(devices[0])); HIP_TRY(hipMemcpyAsync(blobs[i], blobs[0], bytes, hipMemcpyDeviceToDevice, nullptr)); continue; }
        HIP_TRY(hipSetDevice(devices[i]));
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();    // staged copies then
            else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
        HIP_TRY(hipMemcpyPeerAsync(blobs[i], devices[i], blobs[0], devices[0], bytes, nullptr));
    }
    for (int i = 0; i < n; i++) {                                        // both ends of every copy
        bool seen = false;
        for (int k = 0; k < i; k++) if (devices[k] == devices[i]) seen = true;
        if (seen) continue;
        HIP_TRY(hipSetDevice(devices[i]));
        HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipSetDevice(devices[0]));
    return RAISR_HIP_OK;
}

int raisr_hip_set_model(raisr_hip_ctx* c, int pass_index, const float* bank, int hashkeys, int pixel_types,
                        const double qstr[2], const double qcoh[2], int quant_angle)
{
    if (!c || pass_index < 0 || pass_index > 1) return fail(RAISR_HIP_EINVAL, "bad argument");
    const size_t bytes = raisr_hip_model_blob_bytes(hashkeys, pixel_types);
    if (!bytes) return fail(RAISR_HIP_EINVAL, "bad model geometry");
    std::vector<char> host(bytes);
    int rc = raisr_hip_pack_model_blob(host.data(), bank, hashkeys, pixel_types, qstr, qcoh, quant_angle);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    ModelDev& m = c->model[pass_index];
    if (m.blob && m.bytes != bytes) { (void)hipFree(m.blob); m.blob = nullptr; }
    if (!m.blob) { if (hipMalloc(&m.blob, bytes) != hipSuccess) return fail(RAISR_HIP_ENOMEM, "model blob alloc"); }
    HIP_TRY(hipMemcpy(m.blob, host.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());           // frames run on non-blocking streams, which nothing orders after a null-stream copy
    m.bytes = bytes; memcpy(&m.h, host.data(), sizeof m.h); m.valid = true; m.bank_mfma_valid = false;
    fold16_thresholds(m);
    if (int rc = build_lane_major_bank(c, pass_index)) return rc;
    if (int rc = scan_bank_symmetry(c, pass_index, (const float*)(host.data() + kBlobHeader))) return rc;
    return compute_zero_buckets(c, pass_index);
}

static bool fast_mode_supported(const raisr_hip_config* cfg)
{
    return cfg->use_pixel_type && cfg->bits <= 10 && cfg->hash_variant != RAISR_HIP_HASH_FP16;
}

#ifdef RAISR_HIP_DEV
// B panels of k_filter_mfma for every pass in use (590 KB each); filled on the stream by the first frame that needs them
static int alloc_fast_banks(raisr_hip_ctx* c, int passes)
{
    for (int p = 0; p < passes; p++) {
        ModelDev& m = c->model[p];
        if (!m.bank_mfma) {
            if (hipMalloc((void**)&m.bank_mfma, kMfBankHalfs * sizeof(_Float16)) != hipSuccess) { m.bank_mfma = nullptr; return fail(RAISR_HIP_ENOMEM, "fast mode: filter panel alloc"); }
            m.bank_mfma_valid = false;
        }
    }
    return RAISR_HIP_OK;
}

#else
static int alloc_fast_banks(raisr_hip_ctx*, int) { return RAISR_HIP_OK; }
#endif
int raisr_hip_set_fast(raisr_hip_ctx* c, int on)
{
#ifndef RAISR_HIP_DEV
    if (c && on > 0) return fail(RAISR_HIP_EINVAL, kNoFastMode);
#endif
    if (!c) return fail(RAISR_HIP_EINVAL, "null argument");
    if (on && c->configured && !fast_mode_supported(&c->cfg))
        return fail(RAISR_HIP_EINVAL, "fast mode (matrix-core filter stage) supports ratio 2, 8/10-bit content and the fp32 flavours only");
    if (on > 0 && c->configured) {
        HIP_TRY(hipSetDevice(c->device));
        const int rc = alloc_fast_banks(c, c->cfg.passes);
        if (rc) return rc;
    }
    c->fast = on < 0 ? 0 : (on > 2 ? 2 : on);
    return RAISR_HIP_OK;
}

int raisr_hip_get_fast(const raisr_hip_ctx* c) { return c ? c->fast : 0; }

int raisr_hip_configure(raisr_hip_ctx* c, const raisr_hip_config* cfg)
{
    if (!c || !cfg) return fail(RAISR_HIP_EINVAL, "null argument");
    if (cfg->bits != 8 && cfg->bits != 10 && cfg->bits != 16) return fail(RAISR_HIP_EINVAL, "bits must be 8, 10 or 16");
    if (cfg->passes != 1 && cfg->passes != 2) return fail(RAISR_HIP_EINVAL, "passes must be 1 or 2");
    if (cfg->in_width <= 0 || cfg->in_height <= 0 || cfg->out_width <= 0 || cfg->out_height <= 0)
        return fail(RAISR_HIP_EINVAL, "bad plane size");
    if ((uint64_t)cfg->out_width * (uint64_t)cfg->out_height >= (1ull << 31) || (uint64_t)cfg->in_width * (uint64_t)cfg->in_height >= (1ull << 31))
        return fail(RAISR_HIP_EINVAL, "planes of 2^31 samples or more are not supported (32-bit element offsets)");
    if (cfg->clamp_lo < 0 || cfg->clamp_hi <= cfg->clamp_lo || cfg->clamp_hi >= (1 << cfg->bits))
        return fail(RAISR_HIP_EINVAL, "clamp range must satisfy 0 <= lo < hi < 2^bits");
    if (cfg->hash_variant != RAISR_HIP_HASH_AVX2 && cfg->hash_variant != RAISR_HIP_HASH_AVX512 &&
        cfg->hash_variant != RAISR_HIP_HASH_FP16)
        return fail(RAISR_HIP_EINVAL, "unknown hash variant");
    if (cfg->hash_variant == RAISR_HIP_HASH_FP16 && cfg->bits > 10)
        return fail(RAISR_HIP_EINVAL, "the binary16 pipeline supports 8- and 10-bit content (16-bit samples are not exact in binary16)");
    if (cfg->blending != RAISR_HIP_BLEND_COUNT && cfg->blending != RAISR_HIP_BLEND_RANDOMNESS)
        return fail(RAISR_HIP_EINVAL, "blending must be 1 (Randomness) or 2 (CountOfBitsChanged)");
    if (!c->model[0].valid || (cfg->passes == 2 && !c->model[1].valid)) return fail(RAISR_HIP_ESTATE, "model not set");
    for (int p = 0; p < cfg->passes; p++)
        if (c->model[p].h.pixel_types != (cfg->use_pixel_type ? 4 : 1))
            return fail(RAISR_HIP_EINVAL, "model pixel types do not match ratio");
    if (c->sample_shift && cfg->bits + c->sample_shift > (cfg->bits == 8 ? 8 : 16))      // set before this configure: checked against the new sample size
        return fail(RAISR_HIP_EINVAL, "the sample shift set on this context does not fit the configured sample size");
    if (c->fast && !fast_mode_supported(cfg))
        return fail(RAISR_HIP_EINVAL, "fast mode (matrix-core filter stage) supports ratio 2, 8/10-bit content and the fp32 flavours only");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_scratch(c);
    c->configured = false;                     // until every allocation below has succeeded: a failed configure leaves nothing to run on
    c->cfg = *cfg;
    c->gauss = make_gauss(cfg->bits);
    c->sep = make_sep(c->gauss);
    HIP_TRY(hipMemcpy(c->d_gauss, &c->gauss, sizeof(GaussW), hipMemcpyHostToDevice));
    const bool mode2 = cfg->passes == 2 && cfg->two_pass_mode == 2;
    c->passW[0] = mode2 ? cfg->in_width : cfg->out_width;
    c->passH[0] = mode2 ? cfg->in_height : cfg->out_height;
    c->passW[1] = cfg->out_width; c->passH[1] = cfg->out_height;
    const size_t bps = cfg->bits == 8 ? 1 : 2;
    const size_t cap = (size_t)(c->batch_cap < 1 ? 1 : c->batch_cap);     // frame batches: every plane exists `cap` times, back to back
    for (int p = 0; p < cfg->passes; p++) {
        const size_t n = (size_t)c->passW[p] * c->passH[p];
        if (hipMalloc((void**)&c->d_lr[p], cap * n * bps) != hipSuccess ||
            hipMalloc((void**)&c->d_hash[p], cap * n) != hipSuccess ||
            hipMalloc((void**)&c->d_hash2[p], (size_t)c->passH[p] * 16) != hipSuccess ||
            hipMalloc((void**)&c->d_hr[p], cap * n * sizeof(float)) != hipSuccess) {
            free_scratch(c);
            return fail(RAISR_HIP_ENOMEM, "scratch plane alloc");
        }
    }
#ifdef RAISR_HIP_TESTHOOKS
    if (cfg->hash_variant != RAISR_HIP_HASH_FP16) {   // worklists of the split pipeline (largest pass geometry)
        size_t tiles = 0;
        for (int p = 0; p < cfg->passes; p++) {
            const size_t t = (size_t)((c->passW[p] + 63) / 64) * (size_t)((c->passH[p] + 15) / 16);
            if (t > tiles) tiles = t;
        }
        c->fix_tiles = tiles;
        if (hipMalloc((void**)&c->fix.counters, 2 * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&c->fix.counts, tiles * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&c->fix.sparse, tiles * kSparseMax * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&c->fix.dense, tiles * sizeof(unsigned)) != hipSuccess) {
            free_scratch(c);
            return fail(RAISR_HIP_ENOMEM, "worklist alloc");
        }
        HIP_TRY(hipMemsetAsync(c->fix.counters, 0, 2 * sizeof(unsigned), c->stream));
        // list of the deferred exact path (k_hashfilter_ac<.., DEFER> -> k_fix_ac): one region of kWaveCap entries per (tile, wave)
        if (hipMalloc((void**)&c->fixac.counts, cap * tiles * 4) != hipSuccess ||
            hipMalloc((void**)&c->fixac.entries, cap * tiles * 4 * kWaveCap * sizeof(uint16_t)) != hipSuccess) {
            free_scratch(c);
            return fail(RAISR_HIP_ENOMEM, "fix list alloc");
        }
        HIP_TRY(hipMemsetAsync(c->fixac.counts, 0, cap * tiles * 4, c->stream));
    }
#endif
    if (cfg->passes == 2) {
        // pixels the Randomness pass never writes stay 0 in the intermediate (the reference leaves heap garbage there)
        HIP_TRY(hipMemsetAsync(c->d_lr[1], 0, cap * (size_t)c->passW[1] * c->passH[1] * bps, c->stream));
        if (c->passW[0] != c->passW[1] || c->passH[0] != c->passH[1]) {
            const size_t n = (size_t)c->passW[0] * c->passH[0];
            if (hipMalloc((void**)&c->d_mid, cap * n * bps) != hipSuccess) { free_scratch(c); return fail(RAISR_HIP_ENOMEM, "intermediate alloc"); }
            HIP_TRY(hipMemsetAsync(c->d_mid, 0, cap * n * bps, c->stream));
        }
    }
    if (c->fast) {
        const int rc = alloc_fast_banks(c, cfg->passes);
        if (rc) { free_scratch(c); return rc; }
    }
    // The clears above must have LANDED before any frame runs: frames may be enqueued on other (non-blocking) streams -- the
    // ring's, a caller's -- which are not ordered after this one, and a null-stream hipMemset is not ordered with them either
    // (a clear overtaking a frame's pass-1 output showed up as a 1-in-5 mismatch of a small 2-pass test).
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipDeviceSynchronize());
    c->blending = cfg->blending;
    c->configured = true;
    return RAISR_HIP_OK;
}

// MSB-aligned samples of device frames (P010 surfaces: 10-bit values stored as value << 6; VideoDataType::bitShift of the plugin API):
// every device-plane entry point reads sample = stored >> shift and writes stored = sample << shift, as the reference's OpenCL
// pre/post-process kernels do (Raisr_OpenCL_kernel.h:241-276).  Host-plane entry points are not affected.
int raisr_hip_set_sample_shift(raisr_hip_ctx* c, int shift)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    // stored = sample << shift must fit the plane's sample type (8-bit content lives in bytes: no room; 10-bit in 16 bits: up to 6)
    if (shift < 0 || shift > 8 || (shift && c->configured && c->cfg.bits + shift > (c->cfg.bits == 8 ? 8 : 16)))
        return fail(RAISR_HIP_EINVAL, "sample shift out of range for the sample size");
    c->sample_shift = shift;
    return RAISR_HIP_OK;
}

int raisr_hip_set_blending(raisr_hip_ctx* c, int blending)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    if (blending != RAISR_HIP_BLEND_COUNT && blending != RAISR_HIP_BLEND_RANDOMNESS)
        return fail(RAISR_HIP_EINVAL, "blending must be 1 (Randomness) or 2 (CountOfBitsChanged)");
    c->blending = blending;
    return RAISR_HIP_OK;
}

} // extern "C"

// process_y_device with the last pass in `nchunks` row ranges and a callback per finished range (see run_pass)
template <typename RowsDone>
static int process_y_device_impl(raisr_hip_ctx* c, const void* d_in, size_t in_pitch, void* d_out, size_t out_pitch, void* stream,
                                 int nchunks, RowsDone done)
{
    if (!c || !d_in || !d_out) return fail(RAISR_HIP_EINVAL, "null argument");
    if (!c->configured) return fail(RAISR_HIP_ESTATE, "configure first");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const raisr_hip_config& g = c->cfg;
    const int bps = g.bits == 8 ? 1 : 2;
    if (in_pitch % bps || out_pitch % bps) return fail(RAISR_HIP_EINVAL, "pitch not a multiple of the sample size");
    const int ipe = (int)(in_pitch / bps), ope = (int)(out_pitch / bps);

    // Every plane of the pipeline (LR, two-pass intermediate, output) has the sample type of the content:
    // u8 for 8-bit, u16 above.  pass-1 LR = cheap upscale of the input (a copy when pass 1 runs at input size).
    const bool fp16 = g.hash_variant == RAISR_HIP_HASH_FP16;
    const bool same = g.passes == 2 && c->passW[0] == c->passW[1] && c->passH[0] == c->passH[1];
    c->final_out = d_out;
    auto job = [&](auto tag) {
        using T = decltype(tag);
        ResizeParams R0 = make_resize(g.in_width, g.in_height, ipe, c->passW[0], c->passH[0], c->passW[0], g.tie_rule);
        const unsigned nz = (unsigned)c->zb_n;                  // frame batch: one launch per kernel, blockIdx.z = frame
        const size_t plane0 = (size_t)c->passW[0] * c->passH[0], plane1 = (size_t)c->passW[1] * c->passH[1];
        if (nz > 1) { R0.zs_src = c->zb_in_stride; R0.zs_dst = plane0; }
        R0.in_shift = c->sample_shift;                         // MSB-aligned input samples are shifted down on their way into the LR plane
        // pass 1 at input size (two-pass mode 2, Raisr.cpp:960-975) on a tightly pitched plane: the kernels read it where it lies
        c->lr0_alias = (g.in_width == c->passW[0] && g.in_height == c->passH[0] && ipe == c->passW[0] && !c->sample_shift) ? d_in : nullptr;
        if (!c->lr0_alias) launch_resize<T, T>(c, s, d_in, c->d_lr[0], R0, "k_resize", nz);
        if (g.passes == 1) {
            if (fp16) { run_pass16<T>(c, s, 0, d_out, ope, c->zb_out_stride); done(0, g.out_height); } else run_pass<T>(c, s, 0, d_out, ope, nchunks, done, c->zb_out_stride);
            return;
        }
        // pass 1 writes the integer intermediate (Raisr.cpp:927-934).  When both passes run at output size
        // (mode 1) the intermediate IS pass 2's LR plane; in mode 2 it is upscaled now (Raisr.cpp:945-975).
        void* mid = same ? c->d_lr[1] : c->d_mid;
        if (fp16) run_pass16<T>(c, s, 0, mid, c->passW[0], plane0); else run_pass<T>(c, s, 0, mid, c->passW[0], 1, NoRowsDone(), plane0);
        if (!same) {
            ResizeParams R1 = make_resize(c->passW[0], c->passH[0], c->passW[0], c->passW[1], c->passH[1], c->passW[1], g.tie_rule);
            if (nz > 1) { R1.zs_src = plane0; R1.zs_dst = plane1; }
            launch_resize<T, T>(c, s, c->d_mid, c->d_lr[1], R1, "k_resize", nz);
        }
        if (fp16) { run_pass16<T>(c, s, 1, d_out, ope, c->zb_out_stride); done(0, g.out_height); } else run_pass<T>(c, s, 1, d_out, ope, nchunks, done, c->zb_out_stride);
    };
    if (bps == 1) job(uint8_t{}); else job(uint16_t{});
    HIP_TRY(hipGetLastError());
    return RAISR_HIP_OK;
}

extern "C" {

// Frame batch (north_star: "frame batches"): n frames of the configured geometry through ONE launch per kernel, blockIdx.z =
// frame, scratch planes n deep.  Small frames (540p, 720p) leave workgroup slots empty at the head and tail of every launch and
// pay every launch's dispatch gap per frame; a batch pays them once.  One launch per kernel needs the frames equally spaced in
// memory (plane i = plane 0 + i * stride, as in one allocation of n planes) and one of the production pipelines; anything else
// -- and n == 1 -- runs frame by frame with the same results.
int raisr_hip_process_y_device_batch(raisr_hip_ctx* c, int n, const void* const* d_in, size_t in_pitch,
                                     void* const* d_out, size_t out_pitch, void* stream)
{
    if (!c || !d_in || !d_out || n < 1) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (!c->configured) return fail(RAISR_HIP_ESTATE, "configure first");
    for (int i = 0; i < n; i++) if (!d_in[i] || !d_out[i]) return fail(RAISR_HIP_EINVAL, "null plane in the batch");
    const raisr_hip_config& g = c->cfg;
    const int bps = g.bits == 8 ? 1 : 2;
    const bool fp16 = g.hash_variant == RAISR_HIP_HASH_FP16;
    bool one_launch = n > 1 && n <= RAISR_HIP_MAX_BATCH && c->fused && c->blending != RAISR_HIP_BLEND_RANDOMNESS &&
                      (fp16 || (c->certify && !c->split && !c->fast));
    ptrdiff_t si = 0, so = 0;
    if (one_launch) {
        si = (const char*)d_in[1] - (const char*)d_in[0]; so = (char*)d_out[1] - (char*)d_out[0];
        for (int i = 2; i < n && one_launch; i++)
            one_launch = ((const char*)d_in[i] - (const char*)d_in[i - 1]) == si && ((char*)d_out[i] - (char*)d_out[i - 1]) == so;
        // planes of a batch must not overlap: at least one plane apart
        one_launch = one_launch && si >= (ptrdiff_t)(in_pitch * (size_t)g.in_height) && so >= (ptrdiff_t)(out_pitch * (size_t)g.out_height) && si % bps == 0 && so % bps == 0;
    }
    if (!one_launch) {
        for (int i = 0; i < n; i++) {
            const int rc = process_y_device_impl(c, d_in[i], in_pitch, d_out[i], out_pitch, stream, 1, NoRowsDone());
            if (rc) return rc;
        }
        return RAISR_HIP_OK;
    }
    if (n > c->batch_cap) {                                     // grow the scratch planes: a re-configure with the same geometry
        const raisr_hip_config cfg = c->cfg;
        const int blending = c->blending;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : c->stream));
        // (a device-wide synchronise and a reallocation of every scratch plane: once per context and batch size)
        const int old_cap = c->batch_cap;
        c->batch_cap = n;
        const int rc = raisr_hip_configure(c, &cfg);
        c->blending = blending;
        if (rc) { c->batch_cap = old_cap; return rc; }             // the context is unconfigured now (raisr_hip_configure's contract); the cap did not grow
    }
    c->zb_n = n; c->zb_in_stride = (size_t)si / bps; c->zb_out_stride = (size_t)so / bps;
    const int rc = process_y_device_impl(c, d_in[0], in_pitch, d_out[0], out_pitch, stream, 1, NoRowsDone());
    c->zb_n = 1; c->zb_in_stride = c->zb_out_stride = 0;
    return rc;
}

int raisr_hip_process_y_device(raisr_hip_ctx* c, const void* d_in, size_t in_pitch, void* d_out, size_t out_pitch, void* stream)
{
    return process_y_device_impl(c, d_in, in_pitch, d_out, out_pitch, stream, 1, NoRowsDone());
}

int raisr_hip_resize_plane_device(raisr_hip_ctx* c, const void* d_src, int sw, int sh, size_t spitch,
                                  void* d_dst, int dw, int dh, size_t dpitch, int bits, void* stream)
{
    return raisr_hip_resize_plane_device_ex(c, d_src, sw, sh, spitch, 1, d_dst, dw, dh, dpitch, 1, bits, stream);
}

// ... with element steps: sstep / dstep = 2 addresses one channel of an interleaved two-channel plane (NV12 / P010 chroma); the
// pointers then point at the channel's first sample and the widths count samples of that channel
int raisr_hip_resize_plane_device_ex(raisr_hip_ctx* c, const void* d_src, int sw, int sh, size_t spitch, int sstep,
                                     void* d_dst, int dw, int dh, size_t dpitch, int dstep, int bits, void* stream)
{
    if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (sstep < 1 || sstep > 4 || dstep < 1 || dstep > 4) return fail(RAISR_HIP_EINVAL, "element step out of range");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const int bps = bits == 8 ? 1 : 2;
    ResizeParams R = make_resize(sw, sh, (int)(spitch / bps), dw, dh, (int)(dpitch / bps), c->configured ? c->cfg.tie_rule : 0);
    R.sstep = sstep; R.dstep = dstep;
    R.in_shift = R.out_shift = c->sample_shift;
    if (bps == 1) launch_resize<uint8_t, uint8_t>(c, s, d_src, d_dst, R, "k_resize_chroma");
    else launch_resize<uint16_t, uint16_t>(c, s, d_src, d_dst, R, "k_resize_chroma");
    HIP_TRY(hipGetLastError());
    return RAISR_HIP_OK;
}

// Whole device-resident yuv frame (the zero-copy analogue of the reference's vf_raisr_opencl path): RAISR on Y,
// cheap upscale on both chroma planes, all enqueued on `stream`.
int raisr_hip_process_frame_device(raisr_hip_ctx* c,
                                   const void* d_in_y, size_t in_y_pitch, void* d_out_y, size_t out_y_pitch,
                                   const void* d_in_u, const void* d_in_v, size_t in_c_pitch,
                                   void* d_out_u, void* d_out_v, size_t out_c_pitch,
                                   int cin_w, int cin_h, int cout_w, int cout_h, void* stream)
{
    return raisr_hip_process_frame_device_ex(c, d_in_y, in_y_pitch, d_out_y, out_y_pitch, d_in_u, d_in_v, in_c_pitch, d_out_u, d_out_v, out_c_pitch,
                                             cin_w, cin_h, cout_w, cout_h, 1, stream);
}

// chroma_step = 2: U and V are the two channels of ONE interleaved plane (NV12 / P010), on the input and on the output side
int raisr_hip_process_frame_device_ex(raisr_hip_ctx* c,
                                      const void* d_in_y, size_t in_y_pitch, void* d_out_y, size_t out_y_pitch,
                                      const void* d_in_u, const void* d_in_v, size_t in_c_pitch,
                                      void* d_out_u, void* d_out_v, size_t out_c_pitch,
                                      int cin_w, int cin_h, int cout_w, int cout_h, int chroma_step, void* stream)
{
    if (!c || !d_in_u || !d_in_v || !d_out_u || !d_out_v) return fail(RAISR_HIP_EINVAL, "null plane");
    if (!c->configured) return fail(RAISR_HIP_ESTATE, "configure first");
    int rc = raisr_hip_process_y_device(c, d_in_y, in_y_pitch, d_out_y, out_y_pitch, stream);
    if (rc) return rc;
    rc = raisr_hip_resize_plane_device_ex(c, d_in_u, cin_w, cin_h, in_c_pitch, chroma_step, d_out_u, cout_w, cout_h, out_c_pitch, chroma_step, c->cfg.bits, stream);
    if (rc) return rc;
    return raisr_hip_resize_plane_device_ex(c, d_in_v, cin_w, cin_h, in_c_pitch, chroma_step, d_out_v, cout_w, cout_h, out_c_pitch, chroma_step, c->cfg.bits, stream);
}

int raisr_hip_synchronize(raisr_hip_ctx* c)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->bounce.unpack.empty()) HIP_TRY(c->bounce.finish());          // downloads into pageable planes: unpack as their copies complete
    if (c->up) {                               // shared streams: wait for this context's last frame only
        if (c->done_pending) { HIP_TRY(hipEventSynchronize(c->ev_done)); c->done_pending = false; }
        if (c->legacy_pending) {               // a banded frame (rows != NULL) went through the context's own stream pair
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream2));
            c->legacy_pending = false;
        }
        return RAISR_HIP_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    return RAISR_HIP_OK;
}

// Host-plane entry, whole frames: run the last pass in n row ranges and download every finished range while the next one is
// computed (1 = one download after the frame).  Page-locked planes: the copy engine writes them directly; pageable planes: the
// rows of range i are unpacked from the bounce memory while range i+1 is computed (older text: a pageable download blocks the
// calling thread); RAISR_HIP_CHUNKS overrides.
int raisr_hip_set_chunks(raisr_hip_ctx* c, int n)
{
    if (!c || n < 1 || n > 8) return fail(RAISR_HIP_EINVAL, "chunks must be 1..8");
    if (!getenv("RAISR_HIP_CHUNKS")) c->chunks = n;
    return RAISR_HIP_OK;
}

// Order this context's Y kernels (host-plane entry) after those of `prev` (NULL: no ordering).  Both contexts on one device.
int raisr_hip_set_after(raisr_hip_ctx* c, raisr_hip_ctx* prev)
{
    if (!c || c == prev) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (prev && prev->device != c->device) return fail(RAISR_HIP_EINVAL, "contexts on different devices");
    HIP_TRY(hipSetDevice(c->device));
    if (prev && !prev->ev_kern) HIP_TRY(hipEventCreateWithFlags(&prev->ev_kern, hipEventDisableTiming));
    c->after = prev;
    return RAISR_HIP_OK;
}

int raisr_hip_use_streams(raisr_hip_ctx* c, void* compute, void* upload, void* download)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (c->up) { if (c->done_pending) { HIP_TRY(hipEventSynchronize(c->ev_done)); c->done_pending = false; } }
    else { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipStreamSynchronize(c->stream2)); }
    if (!compute && !upload && !download) {    // back to the context's own streams
        if (c->own_stream) { c->stream = c->own_stream; c->own_stream = nullptr; }
        c->up = c->down = nullptr;
        return RAISR_HIP_OK;
    }
    if (!compute || !upload || !download) return fail(RAISR_HIP_EINVAL, "compute, upload and download streams go together");
    if (!c->ev_up) HIP_TRY(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    if (!c->ev_comp) HIP_TRY(hipEventCreateWithFlags(&c->ev_comp, hipEventDisableTiming));
    if (!c->ev_done) HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    if (!c->own_stream) c->own_stream = c->stream;
    c->stream = (hipStream_t)compute; c->up = (hipStream_t)upload; c->down = (hipStream_t)download;
    c->done_pending = false;
    return RAISR_HIP_OK;
}

// Band plan: see include/raisr_hip.h.  Validity of the kept rows (output rows, counted from an ARTIFICIAL
// sub-frame border; a real frame border needs no padding):
//   cheap upscale only      the first/last output rows interpolate against a replicated input row      -> 1 input row
//   one pass                LR row 0 is wrong (replicated input row); HR row q reads LR rows q-6..q+6 and is
//                           unfiltered for q < 6; the blend of row r reads HR rows r-1..r+1           -> r >= 8 output rows
//   two passes, mode 1      pass 2 reads pass-1 rows r-7..r+7, which must be valid themselves          -> r >= 16
//   two passes, mode 2      pass 1 (input size) valid from input row 7, upscaled, then as one pass     -> 12 (2x) / 14 (1.5x) input rows
// The padding below (6 input rows for one pass at 2x, 16 for two, 2 for chroma) covers these with margin; it is
// a multiple of the alignment unit so that band starts keep the upscale phase and the pixel-type parity.
int raisr_hip_plan_bands(int in_height, int out_height, int passes, int nbands, raisr_hip_band* bands)
{
    if (in_height <= 0 || out_height <= 0 || passes < 0 || passes > 2 || nbands < 1 || !bands) return fail(RAISR_HIP_EINVAL, "bad band request");
    const int g = gcd_int(in_height, out_height);
    const int num = out_height / g, den = in_height / g;
    const int align = (den % 2 == 0) ? den : 2 * den;         // band starts: whole upscale periods, even rows
    // input rows of padding at an artificial border: the validity distances above, converted to input rows, plus margin
    const int out8 = (8 * den + num - 1) / num;                // 8 output rows in input rows, rounded up
    int pad = passes == 0 ? 2 : (passes == 1 ? out8 + 2 : 9 + out8 + 3);
    pad = (pad + align - 1) / align * align;
    int K = nbands;
    const int min_rows = 2 * pad + 2 * align;                  // a band keeps at least this many input rows
    if (align > 32 || in_height / min_rows < 2) K = 1;
    else if (K > in_height / min_rows) K = in_height / min_rows;
    for (int k = 0; k < K; k++) {
        const int s0 = k == 0 ? 0 : (int)((long long)in_height * k / K) / align * align;
        const int s1 = k == K - 1 ? in_height : (int)((long long)in_height * (k + 1) / K) / align * align;
        raisr_hip_band& b = bands[k];
        b.in_row_begin = s0 - pad > 0 ? s0 - pad : 0;
        const int in_end = s1 + pad < in_height ? s1 + pad : in_height;
        b.in_row_count = in_end - b.in_row_begin;
        b.out_row_begin = (int)((long long)b.in_row_begin * num / den);
        const int out_end = in_end == in_height ? out_height : (int)((long long)in_end * num / den);
        b.out_row_count = out_end - b.out_row_begin;
        b.keep_begin = (int)((long long)s0 * num / den);
        b.keep_count = (s1 == in_height ? out_height : (int)((long long)s1 * num / den)) - b.keep_begin;
    }
    return K;
}

// 2-D plane copy; contiguous planes (pitch == row bytes on both sides) go as one 1-D copy
static hipError_t copy_plane(void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes, size_t rows,
                             hipMemcpyKind kind, hipStream_t s)
{
    if (dpitch == row_bytes && spitch == row_bytes) return hipMemcpyAsync(dst, src, row_bytes * rows, kind, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, kind, s);
}

// One plane (or row range) between the caller's host memory and the device.  Page-locked host memory: an asynchronous copy.
// Pageable host memory: through the context's bounce memory (host_copy.h) -- packed now (upload) or unpacked when the frame is
// synchronised (download).  The library never hands pageable memory to a HIP copy call: a sporadic GPU page fault of round 3
//  occurred only on the runtime's own pageable-copy paths.  A development build (-DRAISR_HIP_DEV) keeps
// RAISR_HIP_BOUNCE=0 -- pageable planes straight to the runtime, as rounds 1-2 did -- for A/B measurements and fault hunting.
static hipError_t host_copy(raisr_hip_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes, size_t rows,
                            hipMemcpyKind kind, hipStream_t s)
{
    if (!rows || !row_bytes) return hipSuccess;
#ifdef RAISR_HIP_DEV
    static const bool bounce_on = !(getenv("RAISR_HIP_BOUNCE") && atoi(getenv("RAISR_HIP_BOUNCE")) == 0);
#else
    constexpr bool bounce_on = true;
#endif
    const bool h2d = kind == hipMemcpyHostToDevice;
    const void* hp = h2d ? src : dst;
    const size_t hpitch = h2d ? spitch : dpitch;
    if (!bounce_on || HostBounce::page_locked(hp, hpitch * (rows - 1) + row_bytes)) return copy_plane(dst, dpitch, src, spitch, row_bytes, rows, kind, s);
    char* b = c->bounce.take(row_bytes * rows);
    if (!b) return hipErrorOutOfMemory;
    if (h2d) {
        RowCopyPool::get(c->numa_node).copy(b, row_bytes, (const char*)src, spitch, row_bytes, rows);
        hipError_t e = copy_plane(dst, dpitch, b, row_bytes, row_bytes, rows, kind, s);
        if (e != hipSuccess) return e;
        hipEvent_t ev = c->bounce.next_event();                // the bounce memory is reusable once this copy has been executed
        if (!ev) return hipErrorOutOfMemory;
        e = hipEventRecord(ev, s);
        if (e == hipSuccess) c->bounce.uploads.push_back(ev);
        return e;
    }
    hipError_t e = copy_plane(b, row_bytes, src, spitch, row_bytes, rows, kind, s);
    if (e != hipSuccess) return e;
    hipEvent_t ev = c->bounce.next_event();
    if (!ev) return hipErrorOutOfMemory;
    e = hipEventRecord(ev, s);
    if (e != hipSuccess) return e;
    c->bounce.unpack.push_back({(char*)dst, dpitch, b, row_bytes, rows, ev});
    return hipSuccess;
}

int raisr_hip_process_host(raisr_hip_ctx* c,
                           const void* in_y, size_t in_y_pitch, void* out_y, size_t out_y_pitch,
                           const void* in_u, size_t in_u_pitch, void* out_u, size_t out_u_pitch,
                           const void* in_v, size_t in_v_pitch, void* out_v, size_t out_v_pitch,
                           int cin_w, int cin_h, int cout_w, int cout_h)
{
    const int rc = raisr_hip_process_host_async(c, in_y, in_y_pitch, out_y, out_y_pitch, in_u, in_u_pitch, out_u, out_u_pitch,
                                                in_v, in_v_pitch, out_v, out_v_pitch, cin_w, cin_h, cout_w, cout_h, nullptr);
    if (rc) return rc;
    return raisr_hip_synchronize(c);
}

int raisr_hip_process_host_async(raisr_hip_ctx* c,
                                 const void* in_y, size_t in_y_pitch, void* out_y, size_t out_y_pitch,
                                 const void* in_u, size_t in_u_pitch, void* out_u, size_t out_u_pitch,
                                 const void* in_v, size_t in_v_pitch, void* out_v, size_t out_v_pitch,
                                 int cin_w, int cin_h, int cout_w, int cout_h, const raisr_hip_rows* rows)
{
    if (!c || !in_y || !out_y) return fail(RAISR_HIP_EINVAL, "null plane");
    if (!c->configured) return fail(RAISR_HIP_ESTATE, "configure first");
    HIP_TRY(hipSetDevice(c->device));
    // host planes are LSB-aligned whatever raisr_hip_set_sample_shift said for the device-frame entries (raisr_hip.h; the reference's
    // CPU path ignores bitShift): the kernels launched from here run with shift 0, the context's setting comes back on every exit
    struct ShiftOff { raisr_hip_ctx* c; int saved; ~ShiftOff() { c->sample_shift = saved; } } shift_off{c, c->sample_shift};
    c->sample_shift = 0;
    const raisr_hip_config& g = c->cfg;
    const int bps = g.bits == 8 ? 1 : 2;
    const bool chroma = in_u && out_u && in_v && out_v && cin_w > 0 && cin_h > 0 && cout_w > 0 && cout_h > 0;
    const int y_skip = rows ? rows->y_skip : 0, y_keep = rows ? rows->y_keep : g.out_height;
    const int c_skip = rows ? rows->c_skip : 0, c_keep = rows ? rows->c_keep : cout_h;
    const int stage = rows ? rows->stage : 0;
    if (stage < 0 || stage > 2) return fail(RAISR_HIP_EINVAL, "bad stage");
    const bool do_up = stage != 2, do_down = stage != 1;
    if (y_skip < 0 || y_keep < 0 || y_skip + y_keep > g.out_height || (chroma && (c_skip < 0 || c_keep < 0 || c_skip + c_keep > cout_h)))
        return fail(RAISR_HIP_EINVAL, "row window outside the plane");
    // tightly packed device staging: [inY][inU][inV][outY][outU][outV]
    const size_t iy = (size_t)g.in_width * g.in_height * bps, oy = (size_t)g.out_width * g.out_height * bps;
    const size_t ic = chroma ? (size_t)cin_w * cin_h * bps : 0, oc = chroma ? (size_t)cout_w * cout_h * bps : 0;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t off_iu = al(iy), off_iv = off_iu + al(ic), off_oy = off_iv + al(ic), off_ou = off_oy + al(oy), off_ov = off_ou + al(oc);
    const size_t total = off_ov + al(oc);
    if (c->d_stage_bytes < total) {
        pool_put_stage(c->device, c->d_stage, c->d_stage_bytes);
        c->d_stage_bytes = 0;
        c->d_stage = pool_get_stage(c->device, total, &c->d_stage_bytes);
        if (!c->d_stage) return fail(RAISR_HIP_ENOMEM, "staging alloc");
        // the alignment gaps between the staged planes travel to the host with a packed-frame download: never another job's bytes
        HIP_TRY(hipMemsetAsync(c->d_stage, 0, c->d_stage_bytes, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    char* d = (char*)c->d_stage;
    if (do_up) {
        // room for every plane of the frame in the bounce memory (used only by planes that turn out to be pageable)
        if (!c->bounce.unpack.empty()) (void)c->bounce.finish();        // a caller that never synchronised the previous frame
        c->bounce.begin_frame(al(iy) + 2 * al(ic) + 2 * al(oy) + 2 * al(oc) + 16 * 256);
    }
    if (c->up && !rows) {
        // stage-ordered pipeline of the stream ring: uploads on the shared upload stream, kernels (Y, then the two cheap chroma
        // upscales) on the compute stream, one download on the shared download stream; events carry the dependencies on the device
        const size_t irow = (size_t)g.in_width * bps, orow = (size_t)g.out_width * bps;
        const size_t cirow = (size_t)cin_w * bps, crow = (size_t)cout_w * bps;
        hipStream_t s = c->stream;
        HIP_TRY(host_copy(c, d, irow, in_y, in_y_pitch, irow, g.in_height, hipMemcpyHostToDevice, c->up));
        if (c->blending == RAISR_HIP_BLEND_RANDOMNESS)
            HIP_TRY(host_copy(c, d + off_oy, orow, out_y, out_y_pitch, orow, g.out_height, hipMemcpyHostToDevice, c->up));
        if (chroma) {
            HIP_TRY(host_copy(c, d + off_iu, cirow, in_u, in_u_pitch, cirow, cin_h, hipMemcpyHostToDevice, c->up));
            HIP_TRY(host_copy(c, d + off_iv, cirow, in_v, in_v_pitch, cirow, cin_h, hipMemcpyHostToDevice, c->up));
        }
        // (kept also when upload and compute stream are the same: without this marker and the one before the download the same
        //  ring measures 2.3-3.6 k fps instead of 3.9-4.2 k -- the runtime batches the stream's commands differently)
        HIP_TRY(hipEventRecord(c->ev_up, c->up));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_up, 0));
        int rc = raisr_hip_process_y_device(c, d, irow, d + off_oy, orow, s);
        if (rc) return rc;
        if (chroma) {
            rc = raisr_hip_resize_plane_device(c, d + off_iu, cin_w, cin_h, cirow, d + off_ou, cout_w, cout_h, crow, g.bits, s);
            if (rc) return rc;
            rc = raisr_hip_resize_plane_device(c, d + off_iv, cin_w, cin_h, cirow, d + off_ov, cout_w, cout_h, crow, g.bits, s);
            if (rc) return rc;
        }
        if (c->down != s) {                    // (measured: a download behind a cross-stream event, or in a pipeline whose uploads run on a
            HIP_TRY(hipEventRecord(c->ev_comp, s));                     //  shared upload stream, is executed by a copy KERNEL, not the DMA
            HIP_TRY(hipStreamWaitEvent(c->down, c->ev_comp, 0));        //  engine -- the ring passes one stream for all three roles)
        } else {
            HIP_TRY(hipEventRecord(c->ev_comp, c->stream2));            // marker between the last kernel and the download (see above)
            HIP_TRY(hipStreamWaitEvent(s, c->ev_comp, 0));
        }
        const bool packed = chroma && out_y_pitch == orow && out_u_pitch == crow && out_v_pitch == crow &&
                            (const char*)out_u == (const char*)out_y + (off_ou - off_oy) && (const char*)out_v == (const char*)out_y + (off_ov - off_oy);
        if (packed) {
            HIP_TRY(host_copy(c, out_y, (off_ov - off_oy) + oc, d + off_oy, (off_ov - off_oy) + oc, (off_ov - off_oy) + oc, 1, hipMemcpyDeviceToHost, c->down));
        } else {
            HIP_TRY(host_copy(c, out_y, out_y_pitch, d + off_oy, orow, orow, g.out_height, hipMemcpyDeviceToHost, c->down));
            if (chroma) {
                HIP_TRY(host_copy(c, out_u, out_u_pitch, d + off_ou, crow, crow, cout_h, hipMemcpyDeviceToHost, c->down));
                HIP_TRY(host_copy(c, out_v, out_v_pitch, d + off_ov, crow, crow, cout_h, hipMemcpyDeviceToHost, c->down));
            }
        }
        HIP_TRY(hipEventRecord(c->ev_done, c->down));
        c->done_pending = true;
        return RAISR_HIP_OK;
    }
    hipStream_t s = c->stream, s2 = c->stream2;
    if (c->up) c->legacy_pending = true;       // raisr_hip_synchronize must wait for these two streams as well
    // Y: upload, RAISR passes, download on the context stream; chroma (plain cheap upscale, Raisr.cpp:1373-1388)
    // runs on a second stream so its PCIe transfers overlap the Y kernels.
    const size_t irow = (size_t)g.in_width * bps, orow = (size_t)g.out_width * bps;
    const size_t cirow = (size_t)cin_w * bps, crow = (size_t)cout_w * bps;
    bool y_chunked = false;                    // Y rows already on their way back (stage 0 only: upload and download in one call)
    if (do_up) {
        HIP_TRY(host_copy(c, d, irow, in_y, in_y_pitch, irow, g.in_height, hipMemcpyHostToDevice, s));
        if (c->blending == RAISR_HIP_BLEND_RANDOMNESS && y_keep > 0)   // pixels the reference leaves untouched keep the caller's bytes
            HIP_TRY(host_copy(c, d + off_oy + y_skip * orow, orow, out_y, out_y_pitch, orow, y_keep, hipMemcpyHostToDevice, s));
        if (c->after && c->after->ev_kern_valid) HIP_TRY(hipStreamWaitEvent(s, c->after->ev_kern, 0));
        int rc;
        y_chunked = !rows && do_down && c->chunks > 1 && c->blending != RAISR_HIP_BLEND_RANDOMNESS;
        bool chroma_done = false;
        if (y_chunked && chroma) {
            // the second stream carries the Y rows back as they are finished: the (cheap) chroma planes go through it first, their
            // download uses the link while nothing else does
            HIP_TRY(host_copy(c, d + off_iu, cirow, in_u, in_u_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
            HIP_TRY(host_copy(c, d + off_iv, cirow, in_v, in_v_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
            rc = raisr_hip_resize_plane_device(c, d + off_iu, cin_w, cin_h, cirow, d + off_ou, cout_w, cout_h, crow, g.bits, s2);
            if (rc) return rc;
            rc = raisr_hip_resize_plane_device(c, d + off_iv, cin_w, cin_h, cirow, d + off_ov, cout_w, cout_h, crow, g.bits, s2);
            if (rc) return rc;
            if (c_keep > 0) {
                HIP_TRY(host_copy(c, out_u, out_u_pitch, d + off_ou + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
                HIP_TRY(host_copy(c, out_v, out_v_pitch, d + off_ov + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
            }
            chroma_done = true;
        }
        if (y_chunked) {
            // whole frame, synchronous caller: rows leave as they are finished.  The callback runs right after a range's kernels
            // are enqueued on s: an event marks the point, the second stream waits for it and copies the rows back.
            int idx = 0;
            hipError_t cb_err = hipSuccess;
            auto rows_done = [&](int r0, int n) {
                if (cb_err != hipSuccess || n <= 0) return;
                hipEvent_t& ev = c->ev_chunk[idx & 7];
                idx++;
                if (!ev) cb_err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (cb_err == hipSuccess) cb_err = hipEventRecord(ev, s);
                if (cb_err == hipSuccess) cb_err = hipStreamWaitEvent(s2, ev, 0);
                if (cb_err == hipSuccess)
                    cb_err = host_copy(c, (char*)out_y + (size_t)r0 * out_y_pitch, out_y_pitch, d + off_oy + (size_t)r0 * orow, orow, orow, (size_t)n, hipMemcpyDeviceToHost, s2);
            };
            rc = process_y_device_impl(c, d, irow, d + off_oy, orow, s, c->chunks, rows_done);
            if (!rc && cb_err != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "chunked download", cb_err);
        } else {
            rc = raisr_hip_process_y_device(c, d, irow, d + off_oy, orow, s);
        }
        if (rc) return rc;
        if (c->ev_kern) { HIP_TRY(hipEventRecord(c->ev_kern, s)); c->ev_kern_valid = true; }
        if (chroma && !chroma_done) {
            HIP_TRY(host_copy(c, d + off_iu, cirow, in_u, in_u_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
            HIP_TRY(host_copy(c, d + off_iv, cirow, in_v, in_v_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
            rc = raisr_hip_resize_plane_device(c, d + off_iu, cin_w, cin_h, cirow, d + off_ou, cout_w, cout_h, crow, g.bits, s2);
            if (rc) return rc;
            rc = raisr_hip_resize_plane_device(c, d + off_iv, cin_w, cin_h, cirow, d + off_ov, cout_w, cout_h, crow, g.bits, s2);
            if (rc) return rc;
        }
    }
    if (do_down) {
        // Packed output frame: when the caller's three output planes sit in host memory exactly as the staging planes sit in
        // device memory (raisr_hip_packed_frame_layout), the whole frame goes back as ONE copy -- fewer, larger PCIe
        // transfers (the download is what bounds a streamed 4K job: 12.4 MB per frame).
        const bool packed = chroma && !rows && !y_chunked && out_y_pitch == orow && out_u_pitch == crow && out_v_pitch == crow &&
                            (const char*)out_u == (const char*)out_y + (off_ou - off_oy) && (const char*)out_v == (const char*)out_y + (off_ov - off_oy);
        if (packed) {
            if (!c->ev_chroma) HIP_TRY(hipEventCreateWithFlags(&c->ev_chroma, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->ev_chroma, s2));
            HIP_TRY(hipStreamWaitEvent(s, c->ev_chroma, 0));
            HIP_TRY(host_copy(c, out_y, (off_ov - off_oy) + oc, d + off_oy, (off_ov - off_oy) + oc, (off_ov - off_oy) + oc, 1, hipMemcpyDeviceToHost, s));
        } else {
            if (chroma && c_keep > 0 && !y_chunked) {
                HIP_TRY(host_copy(c, out_u, out_u_pitch, d + off_ou + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
                HIP_TRY(host_copy(c, out_v, out_v_pitch, d + off_ov + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
            }
            if (y_keep > 0 && !y_chunked) HIP_TRY(host_copy(c, out_y, out_y_pitch, d + off_oy + y_skip * orow, orow, orow, y_keep, hipMemcpyDeviceToHost, s));
        }
    }
    return RAISR_HIP_OK;
}

// Byte offsets of the Y, U and V planes of a packed frame (tight pitches; each plane starts on a 256-byte boundary) and its
// total size: the host-side layout raisr_hip_process_host* recognises and downloads (uploads) as one copy.
int raisr_hip_packed_frame_layout(int y_w, int y_h, int c_w, int c_h, int bits, size_t offsets[3], size_t* total_bytes)
{
    if (y_w <= 0 || y_h <= 0 || c_w < 0 || c_h < 0 || !offsets || !total_bytes) return fail(RAISR_HIP_EINVAL, "bad argument");
    const size_t bps = bits == 8 ? 1 : 2;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t ny = (size_t)y_w * y_h * bps, nc = (size_t)c_w * c_h * bps;
    offsets[0] = 0; offsets[1] = al(ny); offsets[2] = offsets[1] + al(nc);
    *total_bytes = offsets[2] + nc;
    return RAISR_HIP_OK;
}

#ifdef RAISR_HIP_TESTHOOKS        /* ---- include/raisr_hip_debug.h: not in the product library ---- */
int raisr_hip_debug_keep_stages(raisr_hip_ctx* c, int on)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    c->keep_hash_plane = on != 0;
    return RAISR_HIP_OK;
}

int raisr_hip_debug_read_stage(raisr_hip_ctx* c, int pass_index, uint8_t* hash_out, float* hr_out)
{
    if (!c || pass_index < 0 || pass_index > 1) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (!c->configured || !c->d_hash[pass_index]) return fail(RAISR_HIP_ESTATE, "pass not configured");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = (size_t)c->passW[pass_index] * c->passH[pass_index];
    if (hash_out) HIP_TRY(hipMemcpy(hash_out, c->d_hash[pass_index], n, hipMemcpyDeviceToHost));
    if (hr_out) HIP_TRY(hipMemcpy(hr_out, c->d_hr[pass_index], n * sizeof(float), hipMemcpyDeviceToHost));
    return RAISR_HIP_OK;
}

// Certified hash stage: statistics and self-check (see include/raisr_hip.h).
int raisr_hip_debug_certify(raisr_hip_ctx* c, int collect, int check)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (collect && !c->d_cert_stats) HIP_TRY(hipMalloc((void**)&c->d_cert_stats, 8 * sizeof(unsigned)));
    if (c->d_cert_stats) { HIP_TRY(hipMemsetAsync(c->d_cert_stats, 0, 8 * sizeof(unsigned), c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    if (!collect && c->d_cert_stats) { (void)hipFree(c->d_cert_stats); c->d_cert_stats = nullptr; }
    c->cert_check = check != 0;
    return RAISR_HIP_OK;
}

int raisr_hip_debug_certify_stats(raisr_hip_ctx* c, unsigned out[8])
{
    if (!c || !out) return fail(RAISR_HIP_EINVAL, "null argument");
    if (!c->d_cert_stats) return fail(RAISR_HIP_ESTATE, "statistics are not being collected");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_cert_stats, 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    return RAISR_HIP_OK;
}

// Test hook: the class-1 sign table of the certified hash stage as the context built it (65 536 bytes); ESTATE when the class is off.
int raisr_hip_debug_read_c1tab(raisr_hip_ctx* c, uint8_t* out)
{
    if (!c || !out) return fail(RAISR_HIP_EINVAL, "null argument");
    if (!c->d_c1tab) return fail(RAISR_HIP_ESTATE, "the class-1 rule is switched off (RAISR_HIP_C1=0)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(out, c->d_c1tab, kC1Buckets, hipMemcpyDeviceToHost));
    return RAISR_HIP_OK;
}

// Test hook: decision of the certified hash stage for n host-side approximate tensor triples (see include/raisr_hip.h).
int raisr_hip_debug_approx_hash(raisr_hip_ctx* c, int pass_index, int hash_flavour, const float* abd, size_t n,
                                uint8_t* bucket_out, uint8_t* cert_out, float* eps_out)
{
    if (!c || pass_index < 0 || pass_index > 1 || !abd || !bucket_out || !cert_out) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (hash_flavour != RAISR_HIP_HASH_AVX512 && hash_flavour != RAISR_HIP_HASH_AVX2) return fail(RAISR_HIP_EINVAL, "hash_flavour must be AVX512 or AVX2");
    if (!c->model[pass_index].blob || !c->configured) return fail(RAISR_HIP_ESTATE, "set the model and configure first");
    if (eps_out) *eps_out = c->sep.eEb * 2.0f;             // the eps of the tensor bounds (eEb = eps / 2)
    if (n == 0) return RAISR_HIP_OK;
    if (n > 0x7fffffffu / 3) return fail(RAISR_HIP_EINVAL, "too many triples");
    HIP_TRY(hipSetDevice(c->device));
    float* d_in = nullptr; uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, n * 3 * sizeof(float)));
    if (hipMalloc((void**)&d_out, 2 * n) != hipSuccess) { (void)hipFree(d_in); return fail(RAISR_HIP_ENOMEM, "hipMalloc"); }
    int rc = RAISR_HIP_OK;
    PassParams P = make_pass(c, pass_index, 0, 0);
    if (hipMemcpy(d_in, abd, n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipDeviceSynchronize");
    if (!rc) {
        hipLaunchKernelGGL(k_debug_approx_hash, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, (unsigned)n, P, c->sep,
                           hash_flavour == RAISR_HIP_HASH_AVX2 ? 1 : 0, d_out, d_out + n);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "k_debug_approx_hash");
    }
    if (!rc && (hipMemcpy(bucket_out, d_out, n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(cert_out, d_out + n, n, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}

#ifdef RAISR_HIP_DEV
// development builds: wave-cycles per phase of k_hashfilter_ac since the last call (kernels_common.h g_phase_cycles); clears them
int raisr_hip_dev_phase_stats(unsigned long long out[8])
{
    if (hipDeviceSynchronize() != hipSuccess) return fail(RAISR_HIP_ERUNTIME, "hipDeviceSynchronize");
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return fail(RAISR_HIP_ERUNTIME, "hipMemcpyFromSymbol");
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero) != hipSuccess) return fail(RAISR_HIP_ERUNTIME, "hipMemcpyToSymbol");
    return RAISR_HIP_OK;
}
#endif

// Test hook: exhaustive comparison of the binary16 hash's folded thresholds (Pass16) with the divisions they replace, for
// pass `pass_index`'s model.  out[0] = disagreements (must be 0), out[1] = operand pairs compared (~2^31).
int raisr_hip_debug_fold16_check(raisr_hip_ctx* c, int pass_index, unsigned long long out[2])
{
    if (!c || pass_index < 0 || pass_index > 1 || !out) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (!c->model[pass_index].blob) return fail(RAISR_HIP_ESTATE, "model not set for this pass");
    if (!c->model[pass_index].fold16) return fail(RAISR_HIP_ESTATE, "this model's thresholds are not folded (not positive finite numbers)");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, 2 * sizeof(unsigned long long)));
    int rc = RAISR_HIP_OK;
    if (hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), c->stream) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemset");
    if (!rc) {
        const Pass16 Q = make_pass16(c, pass_index, 64);
        hipLaunchKernelGGL(k_debug_fold16, dim3(65536), dim3(256), 0, c->stream, Q, d_out);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "k_debug_fold16");
    }
    if (!rc && hipMemcpy(out, d_out, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    (void)hipFree(d_out);
    return rc;
}

// Test hook: hash bucket of n host-side (a, b, d) triples with pass `pass_index`'s thresholds, computed by the
// device functions k_hash uses (hash_flavour: RAISR_HIP_HASH_AVX512 or RAISR_HIP_HASH_AVX2).
int raisr_hip_debug_hash(raisr_hip_ctx* c, int pass_index, int hash_flavour, const float* abd, size_t n, uint8_t* hash_out)
{
    if (!c || pass_index < 0 || pass_index > 1 || !abd || !hash_out) return fail(RAISR_HIP_EINVAL, "bad argument");
    if (hash_flavour != RAISR_HIP_HASH_AVX512 && hash_flavour != RAISR_HIP_HASH_AVX2) return fail(RAISR_HIP_EINVAL, "hash_flavour must be AVX512 or AVX2");
    if (!c->model[pass_index].blob) return fail(RAISR_HIP_ESTATE, "model not set for this pass");
    if (n == 0) return RAISR_HIP_OK;
    if (n > 0x7fffffffu / 3) return fail(RAISR_HIP_EINVAL, "too many triples");
    HIP_TRY(hipSetDevice(c->device));
    float* d_in = nullptr; uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, n * 3 * sizeof(float)));
    if (hipMalloc((void**)&d_out, n) != hipSuccess) { (void)hipFree(d_in); return fail(RAISR_HIP_ENOMEM, "hipMalloc"); }
    int rc = RAISR_HIP_OK;
    PassParams P = make_pass(c, pass_index, 0, 0);
    if (hipMemcpy(d_in, abd, n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipDeviceSynchronize");
    if (!rc) {
        hipLaunchKernelGGL(k_debug_hash, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, (unsigned)n, P,
                           hash_flavour == RAISR_HIP_HASH_AVX2 ? 1 : 0, d_out);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "k_debug_hash");
    }
    if (!rc && hipMemcpy(hash_out, d_out, n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RAISR_HIP_ERUNTIME, "hipMemcpy");
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}

#endif  /* RAISR_HIP_TESTHOOKS */

// Enable/disable per-kernel HIP-event timing of subsequent process calls (events are recorded on the
// stream each kernel is launched on).
int raisr_hip_kernel_timing_enable(raisr_hip_ctx* c, int on)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    KernelTimer& T = c->timer;
    T.recs.clear(); T.names.clear();
    if (on && T.pool.empty()) {
        HIP_TRY(hipSetDevice(c->device));
        T.pool.resize(2 * T.cap);
        for (auto& e : T.pool) HIP_TRY(hipEventCreate(&e));
    }
    T.enabled = on != 0;
    return RAISR_HIP_OK;
}

// Collects the timings recorded since the last enable: per kernel name, total milliseconds and
// launch count.  Caller must have synchronised the stream(s).  Returns the number of kernels.
int raisr_hip_kernel_timing_read(raisr_hip_ctx* c, char* names_out, float* total_ms_out, int* count_out, int max_kernels)
{
    if (!c || !names_out || !total_ms_out || !count_out) return fail(RAISR_HIP_EINVAL, "null argument");
    KernelTimer& T = c->timer;
    const int n = (int)T.names.size() < max_kernels ? (int)T.names.size() : max_kernels;
    for (int i = 0; i < n; i++) {
        snprintf(names_out + 64 * i, 64, "%s", T.names[i].c_str());
        total_ms_out[i] = 0.f; count_out[i] = 0;
    }
    for (auto& r : T.recs) {
        if (r.id >= n) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms_out[r.id] += ms; count_out[r.id]++; }
    }
    return n;
}

#ifdef RAISR_HIP_TESTHOOKS
int raisr_hip_profile_kernels(raisr_hip_ctx* c, const void* d_in, size_t in_pitch, void* d_out, size_t out_pitch,
                              int iters, char* names_out, float* ms_out, int max_kernels)
{
    if (!c || iters <= 0 || !names_out || !ms_out) return fail(RAISR_HIP_EINVAL, "bad argument");
    int rc = raisr_hip_kernel_timing_enable(c, 1);
    if (rc) return rc;
    for (int i = 0; i < iters; i++) {
        rc = raisr_hip_process_y_device(c, d_in, in_pitch, d_out, out_pitch, nullptr);
        if (rc) return rc;
    }
    rc = raisr_hip_synchronize(c);
    if (rc) return rc;
    std::vector<int> counts(max_kernels);
    int n = raisr_hip_kernel_timing_read(c, names_out, ms_out, counts.data(), max_kernels);
    for (int i = 0; i < n; i++) if (counts[i]) ms_out[i] /= counts[i];
    raisr_hip_kernel_timing_enable(c, 0);
    return n;
}

#endif  /* RAISR_HIP_TESTHOOKS */

}  // extern "C"

