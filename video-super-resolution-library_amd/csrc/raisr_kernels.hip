// raisr_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the Enhanced-RAISR Y-plane hot path
// and the C ABI declared in include/raisr_hip.h.
//
// Pipeline per RAISR pass (whole-frame semantics of the reference's processSegment(),
// Library/Raisr.cpp:890-1289, run with threadcount=1):
//
//   k_resize      cheap upscale (stand-in for ippiResizeLinear, Raisr.cpp:947-958)       in -> LR (sample type)
//   k_hashfilter  per 64 x 16 tile, one launch, two stages sharing one LR window in LDS:
//                   hash stage    11x11 structure tensor + hash (Raisr_AVX512.cpp:69-131,175-258; tail columns
//                                 also Raisr_AVX256.cpp:393-472)                        LR -> bucket per pixel (LDS)
//                   filter stage  hash-indexed 121-tap filter + accept test
//                                 (Raisr_AVX512.cpp:134-149, Raisr.cpp:1196-1200)       LR, bucket -> HR (f32)
//   k_blend       census-transform blend, clamp, narrow, borders
//                 (Raisr_AVX256.cpp:68-166, Raisr.cpp:999-1028,1252-1265)               LR, HR -> out
//   (k_hash + k_filter are the same two stages as separate launches with the bucket plane in HBM: RAISR_HIP_FUSED=0.)
//
// Numeric contract: every floating-point operation below maps to exactly one IEEE-754 binary32
// operation of the cited reference lines ("strict source" semantics).  This file MUST be built
// with -ffp-contract=off and without fast-math; FMAs appear only where the reference has an
// explicit fmadd intrinsic.
//
// (raisr_fp16_kernels.h holds the binary16 twins k_hashfilter16 / k_hash16 / k_filter16 / k_blend16 for the
//  AVX512-FP16 numerics; k_blend_rand is the Randomness blending mode shared by both.)
//
// Design notes (MI355X): the work is fp32-VALU bound (~1.3 kFLOP per output pixel per pass against
// ~1.25 compulsory HBM bytes) and, in the filter stage, vector-L1 bound (512 B of coefficients per pixel),
// so the kernels are organised around VALU/LDS/L1 efficiency:
//   * hash stage: one wave = 64 adjacent columns x 4 rows; gradients are computed once per tile into
//     LDS as (gx,gy) float2 so the inner loop is ds_read_b64 + v_pk_mul_f32 + v_pk_fma_f32 +
//     v_fmac_f32 per (pixel, tap) with the Gaussian weight in an SGPR; column accumulators are
//     folded in the reference's reduction-tree order as they complete; the hash itself is straight-line code.
//   * filter stage: 16 lanes per pixel (one lane per accumulator lane of the reference's zmm), so a
//     filter row is fetched as fully coalesced 64-byte segments and the 16->1 tree is DPP row rotations and
//     quad permutes -- exactly the reference's sumitup_ps_512 association.
//   * both stages in one kernel: workgroups (and waves) of a CU are in different stages at any time, so the VALU-bound
//     and the L1-bound stage overlap on every CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/raisr_hip.h"
#include "x86_approx_tables.h"
#include "x86_approx_dev.h"
#include "x86_fp16_tables.h"

#if defined(__FAST_MATH__)
#error "raisr_kernels.hip must be compiled without fast-math"
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kTaps = 121;
constexpr int kTapsPad = 128;
constexpr int kMargin = 6;       // gLoopMargin, Raisr.cpp:1574
constexpr int kBlobHeader = 64;  // bytes

struct GaussW {
    float wT[11][12];            // wT[k][i] = weight of patch row i, column k (column-major for the k-outer loop)
};

struct PassParams {
    int W, H;                    // plane size of this pass
    int lr_pitch;                // LR plane pitch in u16 elements
    int hash_pitch;              // hash plane pitch in bytes (u8 elements)
    int hr_pitch;                // HR plane pitch in floats
    float lo, hi;                // accept-test / clamp limits as float
    int ilo, ihi;
    int a_begin, a_end;          // columns hashed with the AVX-512 flavour
    int b_begin, b_end;          // columns hashed with the AVX2 flavour (after the AVX-512 one)
    int c_final;                 // first column that is never filtered
    int ov_begin, ov_end;        // columns hashed twice (AVX-512 flavour, then AVX2): second hash lives in hash2
    const uint8_t* hash2;        // [H][16] second hash of the overlap columns (column c -> c - ov_begin)
    int pixel_types;             // 4 (ratio 2) or 1
    int randomness;              // 1: BlendingMode Randomness (tail re-hash candidate replaces, never keeps, the first)
    float qangle, qs0, qs1, qc0, qc1;
    const float* bank;           // [hash][type][128]
    int bank_bytes;              // size of the fp32 bank (buffer-descriptor range)
    const uint2* tab14;          // [128]: rcp14 {C0,C1}[64], rsqrt14 {C0,C1}[64]
    const uint16_t* lut_legacy;  // rcp[2048], rsqrt[2048]
    int write_hash;              // fused kernel: also write the hash plane (introspection for tests)
    unsigned* cert_stats;        // certified-hash kernel: {pixels sent to the exact path, certified-but-wrong, zone pixels} or null
    int cert_check;              // 1: every pixel also takes the exact path and certified buckets are compared with it (tests)
    int zero_bucket[2];          // bucket of the all-zero tensor in the AVX-512 / AVX2 flavour (flat windows), from the exact device code
    const float* gauss_dev;      // GaussW::wT as a device array [11][12] (per-lane weights of the 16-lane exact tensor)
};

// XCD-aware tile order.  The dispatcher hands workgroup b to XCD b % 8 (observed, MI355X_MICROARCH.md) and
// each XCD has a private 4 MiB L2, so with the plain (blockIdx.x, blockIdx.y) order the eight tiles around
// any tile live in eight different L2s and every halo row/column is fetched from HBM again.  Remap the
// dispatch index so that each XCD walks one contiguous row-major strip of tiles: neighbouring tiles then
// share an L2 and the halo re-reads hit it.  Pure performance: any placement gives the same result.
__device__ __forceinline__ void xcd_tile(int& bx, int& by)
{
    const unsigned gx = gridDim.x, n = gridDim.x * gridDim.y;
    const unsigned b = blockIdx.y * gx + blockIdx.x;
    const unsigned n8 = n & ~7u;
    const unsigned t = b < n8 ? (b & 7u) * (n8 >> 3) + (b >> 3) : b;
    by = (int)(t / gx);
    bx = (int)(t - (unsigned)by * gx);
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier, and on
// gfx950 the fence makes every wave wait for ALL its outstanding global loads and stores (s_waitcnt vmcnt(0)) -- which
// turns a register prefetch of the next tile into a stall, and makes a persistent workgroup wait for the write
// acknowledgement of its output stores at every tile.  The kernels below exchange nothing through global memory inside a
// workgroup, so they wait for their LDS traffic only.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// k_resize: centre-aligned bilinear with replicate border in exact integer arithmetic.
// dst(y,x): n = (2d+1)*S - D, den = 2D per axis (reduced by gcd on the host: Sx,Dx,Sy,Dy).
// ------------------------------------------------------------------------------------------------
struct ResizeParams {
    int sw, sh, dw, dh;
    int spitch, dpitch;          // in elements
    int Sx, Dx, Sy, Dy;          // reduced ratios
    int tie_even;
};

__device__ __forceinline__ void axis_tap(int d, int S, int D, int size, int& i0, int& i1, int& f)
{
    const int n = (2 * d + 1) * S - D, den = 2 * D;
    int q = n >= 0 ? n / den : -((-n + den - 1) / den);
    f = n - q * den;
    i0 = min(max(q, 0), size - 1);
    i1 = min(max(q + 1, 0), size - 1);
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_resize(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= R.dw || y >= R.dh) return;
    int x0, x1, fx, y0, y1, fy;
    axis_tap(x, R.Sx, R.Dx, R.sw, x0, x1, fx);
    axis_tap(y, R.Sy, R.Dy, R.sh, y0, y1, fy);
    const long long denx = 2 * R.Dx, deny = 2 * R.Dy;
    const TIn* r0 = src + (size_t)y0 * R.spitch;
    const TIn* r1 = src + (size_t)y1 * R.spitch;
    const long long top = (denx - fx) * (long long)r0[x0] + (long long)fx * r0[x1];
    const long long bot = (denx - fx) * (long long)r1[x0] + (long long)fx * r1[x1];
    const long long num = (deny - fy) * top + fy * bot;
    const long long den = denx * deny;
    long long q = (2 * num + den) / (2 * den);
    if (R.tie_even && ((2 * num + den) % (2 * den) == 0) && (q & 1)) q--;
    dst[(size_t)y * R.dpitch + x] = (TOut)q;
}

// 2x special case of the same arithmetic: weights {1,3}/4 per axis, out = (sum + 8) >> 4
// (ties: +8 then >>4 is round-half-up; the half-even switch subtracts one when the discarded
// bits are exactly 8 and the quotient is odd).  One thread produces 4 adjacent output pixels.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_resize2x(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    int bx, by;
    xcd_tile(bx, by);                                         // vertically adjacent blocks share input rows: keep them on one XCD
    const int t = bx * 64 + (threadIdx.x & 63);               // group of 4 output columns
    const int y = by * 4 + (threadIdx.x >> 6);
    const int x0 = 4 * t;
    if (x0 >= R.dw || y >= R.dh) return;
    const int sy = y >> 1;
    const int ya = (y & 1) ? sy : max(sy - 1, 0);             // far/near rows: even y -> (sy-1: 1, sy: 3)
    const int yb = (y & 1) ? min(sy + 1, R.sh - 1) : sy;      //                odd  y -> (sy: 3, sy+1: 1)
    const int wa = (y & 1) ? 3 : 1, wb = 4 - wa;
    const TIn* ra = src + (size_t)ya * R.spitch;
    const TIn* rb = src + (size_t)yb * R.spitch;
    const int c = 2 * t;                                      // source columns c-1 .. c+2
    const int cm = max(c - 1, 0), c1 = min(c + 1, R.sw - 1), c2 = min(c + 2, R.sw - 1), cc = min(c, R.sw - 1);
    const int a0 = ra[cm], a1 = ra[cc], a2 = ra[c1], a3 = ra[c2];
    const int b0 = rb[cm], b1 = rb[cc], b2 = rb[c1], b3 = rb[c2];
    const int v0 = wa * a0 + wb * b0, v1 = wa * a1 + wb * b1, v2 = wa * a2 + wb * b2, v3 = wa * a3 + wb * b3;
    int o[4] = {v0 + 3 * v1, 3 * v1 + v2, v1 + 3 * v2, 3 * v2 + v3};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int q = (o[i] + 8) >> 4;
        if (R.tie_even && ((o[i] & 15) == 8) && (q & 1)) q--;
        o[i] = q;
    }
    TOut* d = dst + (size_t)y * R.dpitch + x0;
    if (x0 + 3 < R.dw) {
        if (sizeof(TOut) == 2 && ((reinterpret_cast<uintptr_t>(d) & 7) == 0)) {
            *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
        } else if (sizeof(TOut) == 1 && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
            *reinterpret_cast<unsigned*>(d) = (unsigned)o[0] | ((unsigned)o[1] << 8) | ((unsigned)o[2] << 16) | ((unsigned)o[3] << 24);
        } else {
            d[0] = (TOut)o[0]; d[1] = (TOut)o[1]; d[2] = (TOut)o[2]; d[3] = (TOut)o[3];
        }
    } else {
        for (int i = 0; i < 4 && x0 + i < R.dw; i++) d[i] = (TOut)o[i];
    }
}

// same-size case of the cheap upscale (two-pass mode 2 runs pass 1 at input size, Raisr.cpp:960-975): widen/copy
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_copy(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < R.dw && y < R.dh) dst[(size_t)y * R.dpitch + x] = (TOut)src[(size_t)y * R.spitch + x];
}

// ------------------------------------------------------------------------------------------------
// Stage an LH x LWLOAD window of a plane -- origin (y0, x0), replicate-clamped to the W x H plane --
// into an LDS tile of row stride LSTRIDE, converting each sample to TL.  Block = 256 threads, linear sweep
// (every lane busy).  All of a thread's global loads are issued before the first one is consumed: the
// rolled form waited for each load in turn, i.e. ~9 dependent memory round trips at the head of every
// workgroup.
// ------------------------------------------------------------------------------------------------
template <int LH, int LWLOAD, typename T>
struct TileRegs {
    static constexpr int REM = LWLOAD - 64;                    // halo columns right of the first 64
    static constexpr int NM = (LH + 3) / 4;                    // rows per wave in the main part
    static constexpr unsigned NR = LH * REM, NRL = (NR + 255u) / 256u;
    T vm[NM];
    T vr[NRL];
};

// global -> registers half of stage_tile (all loads in flight, nothing waits): lets a persistent workgroup fetch the
// next tile while it computes the current one
template <int LH, int LWLOAD, typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ src, int pitch, int W, int H, int y0, int x0, TileRegs<LH, LWLOAD, T>& R)
{
    static_assert(LWLOAD > 64 && LWLOAD <= 128, "a tile row is one 64-lane sweep plus a remainder");
    using TR = TileRegs<LH, LWLOAD, T>;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: the row arithmetic stays scalar
    __builtin_assume(w >= 0 && w < 4);
    // main part: wave w sweeps columns [0,64) of rows w, w+4, ...: one clamped column index per lane for all rows
    const int gxm = min(max(x0 + lane, 0), W - 1);
#pragma unroll
    for (int it = 0; it < TR::NM; it++) {
        const int gy = min(max(y0 + min(w + 4 * it, LH - 1), 0), H - 1);
        R.vm[it] = src[(unsigned)gy * (unsigned)pitch + (unsigned)gxm];          // planes are < 2^31 samples: 32-bit offsets
    }
    // remainder: the REM right-hand columns of all rows, spread linearly over the block
#pragma unroll
    for (unsigned it = 0; it < TR::NRL; it++) {
        const unsigned idx = min(threadIdx.x + 256u * it, TR::NR - 1u);
        const int ty = (int)(idx / TR::REM), tx = 64 + (int)(idx - (unsigned)ty * TR::REM);
        const int gy = min(max(y0 + ty, 0), H - 1), gx = min(max(x0 + tx, 0), W - 1);
        R.vr[it] = src[(unsigned)gy * (unsigned)pitch + (unsigned)gx];
    }
}

// registers -> LDS half (converting each sample to TL)
template <int LH, int LWLOAD, int LSTRIDE, typename TL, typename T>
__device__ __forceinline__ void store_tile(const TileRegs<LH, LWLOAD, T>& R, TL* sL)
{
    using TR = TileRegs<LH, LWLOAD, T>;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(w >= 0 && w < 4);
#pragma unroll
    for (int it = 0; it < TR::NM; it++) {
        const int ty = w + 4 * it;
        if (ty < LH) sL[ty * LSTRIDE + lane] = (TL)(float)R.vm[it];
    }
#pragma unroll
    for (unsigned it = 0; it < TR::NRL; it++) {
        const unsigned idx = threadIdx.x + 256u * it;
        const int ty = (int)(idx / TR::REM), tx = 64 + (int)(idx - (unsigned)ty * TR::REM);
        if (idx < TR::NR) sL[ty * LSTRIDE + tx] = (TL)(float)R.vr[it];
    }
}

template <int LH, int LWLOAD, int LSTRIDE, typename TL, typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, int pitch, int W, int H, int y0, int x0, TL* sL)
{
    TileRegs<LH, LWLOAD, T> R;
    load_tile<LH, LWLOAD>(src, pitch, W, H, y0, x0, R);
    store_tile<LH, LWLOAD, LSTRIDE>(R, sL);
}

// ------------------------------------------------------------------------------------------------
// hash (per pixel), strict operation order of GetHashValue_AVX512_32f_16Elements
// (Raisr_AVX512.cpp:175-258) / GetHashValue_AVX256_32f_8Elements (Raisr_AVX256.cpp:393-472)
// ------------------------------------------------------------------------------------------------
// sqrt14(v) = VRCP14(VRSQRT14(v)).  Branch-free for everything the hash actually produces:
//   v normal, positive, finite -> both table evaluations stay in the normal range (the rsqrt14 result
//                                 has a biased exponent in [62,190]);
//   v == +-0                   -> rcp14(+-inf) = +-0;
//   v negative (not NaN)       -> rsqrt14 gives the QNaN indefinite, rcp14 passes it through.
// Positive denormals, +inf and NaN inputs (never produced by 8/10-bit content) take the
// generic models of x86_approx_dev.h behind a rarely-taken branch.
// Fully branch-free: inputs outside the three cases above set `rare` and the caller recomputes that
// pixel with the generic models (x86_approx_dev.h) -- so the hashes of a lane's pixels are straight-line
// code the scheduler can interleave.
__device__ __forceinline__ float sqrt14_fast(float v, const uint2* tab, bool& rare)
{
    const uint32_t x = __float_as_uint(v);
    // rsqrt14 table row = [exponent parity p][top 5 mantissa bits] = bits 23..18 of x with bit 23 inverted
    // (p = (E-127)&1 = ~E&1); every bit pattern yields an in-range row, so nothing has to be sanitised first
    uint2 c = tab[64u + (((x >> 18) & 63u) ^ 32u)];
    uint32_t code = (c.x - __umul24(c.y, (x >> 8) & 1023u)) >> 9;          // rsqrt14 mantissa code (16 bits); C1*t < 2^26
    const bool pow4 = (x & 0x00ffffffu) == 0x00800000u;                     // p == 0 && m == 0: exact power of four
    code = pow4 ? 0u : code;
    // rcp14 of the (normal) intermediate y = 2^(-half-1) * (1 + code/65536)  [or 2^-half when pow4]:
    // its top 6 / next 10 mantissa bits are code>>10 / code&1023, so y never has to be assembled
    c = tab[code >> 10];
    uint32_t code2 = (c.x - __umul24(c.y, code & 1023u)) >> 9;
    asm volatile("" : "+v"(code2));     // keep the second look-up unconditional: a branch around it would serialise the lane's 12 roots
    const uint32_t ez = ((x + 0x3f800000u) >> 1) & 0x7f800000u;             // biased exponent (E+127)>>1 of the root
    // y an exact power of two (code == 0): rcp14 is exact as well -- 2^half for a power of four, else one binade up
    const uint32_t zp = ez + (pow4 ? 0u : 0x00800000u);
    uint32_t z = (code == 0u) ? zp : (ez | (code2 << 7));
    const bool normal = __builtin_amdgcn_classf(v, 0x100);               // +normal
    const bool zero = __builtin_amdgcn_classf(v, 0x060);                 // +-0 -> rcp14(+-inf) = +-0
    // -inf, -normal, -denormal -> QNaN indefinite; sNaN, qNaN, +denormal, +inf -> generic model (caller)
    z = normal ? z : (zero ? x : 0xffc00000u);
    rare |= __builtin_amdgcn_classf(v, 0x283);
    return __uint_as_float(z);
}

// RCPPS(RSQRTPS(v)), branch-free (the composition of x86dev::rsqrt_legacy and x86dev::rcp_legacy):
//   +normal: y = RSQRTPS(v) has exponent 126-half and the 12-bit mantissa code q = lut[2048 + 1024p + (m>>13)];
//            RCPPS(y) looks up y's top 11 mantissa bits (= q>>1): exponent 253-(126-half) = (E+127)>>1, mantissa lut[q>>1]<<11;
//   +-0 and +-denormal (DAZ) -> RSQRTPS = +-inf -> RCPPS = +-0;   -normal, -inf -> QNaN indefinite;
//   +inf -> RSQRTPS = +0 -> RCPPS = +inf;   NaN -> quieted NaN.
__device__ __forceinline__ float sqrt_legacy_fast(float v, const uint16_t* lut)
{
    const uint32_t x = __float_as_uint(v);
    const uint32_t q = lut[2048u + (((x >> 13) & 2047u) ^ 1024u)];          // [p = ~E&1][m>>13]: bits 23..13, bit 23 inverted
    const uint32_t r = lut[q >> 1];
    const uint32_t ez = ((x + 0x3f800000u) >> 1) & 0x7f800000u;             // biased exponent (E+127)>>1 of the root
    uint32_t z = ez | (r << 11);
    const bool normal = __builtin_amdgcn_classf(v, 0x100);
    const bool tiny = __builtin_amdgcn_classf(v, 0x0f0);                    // +-0, +-denormal
    const bool negative = __builtin_amdgcn_classf(v, 0x00c);                // -inf, -normal
    const bool nan = __builtin_amdgcn_classf(v, 0x003);
    uint32_t sp = nan ? (x | 0x00400000u) : x;                              // NaN quieted; +inf stays +inf
    sp = negative ? 0xffc00000u : sp;
    sp = tiny ? (x & 0x80000000u) : sp;
    z = normal ? z : sp;
    return __uint_as_float(z);
}

// Hash thresholds, passed BY VALUE (SGPRs): taking PassParams by reference in an out-of-line function would
// force the whole struct into scratch memory.
struct HashQ { float qangle, qs0, qs1, qc0, qc1; const uint16_t* lut; };

// MODE 0: AVX-512 flavour, branch-free fast path (sets `rare` when the generic model is needed);
// MODE 1: AVX2 flavour (legacy LUT instructions); MODE 2: AVX-512 flavour, generic models.
template <int MODE>
__device__ __forceinline__ float sqrt_approx(float v, const uint2* tab, const uint16_t* lut, bool& rare)
{
    if (MODE == 1) return sqrt_legacy_fast(v, lut);
    if (MODE == 2) return x86dev::rcp14(x86dev::rsqrt14(v, tab + 64), tab);
    return sqrt14_fast(v, tab, rare);
}

template <int MODE>
__device__ __forceinline__ int hash_px_impl(float a, float b, float d, const HashQ P, const uint2* tab, bool& rare)
{
    constexpr bool LEGACY = MODE == 1;
    const float pi = 3.141592653f;                       // Raisr_globals.h:29
    const float T = a + d;
    const float Dt = (a * d) - (b * b);
    const float rad = ((T * T) * 0.25f) - Dt;            // x/4 == x*0.25 exactly
    const float s = sqrt_approx<MODE>(rad, tab, P.lut, rare);
    const float hT = T * 0.5f;                           // T/2
    const float L1 = hT + s;
    const float L2 = hT - s;
    const float xx = (b < 0.0f || b > 0.0f) ? (L1 - d) : 1.0f;   // _CMP_NEQ_OQ
    // atan2Approximation (Raisr_AVX512.cpp:151-173)
    const float ONEQTR_PI = (float)(3.14159265358979323846 / 4.0);          // (float)(M_PI/4.0)
    const float THRQTR_PI = (float)(3.0 * 3.14159265358979323846 / 4.0);    // (float)(3.0*M_PI/4.0)
    const float ay = __builtin_fabsf(b) + 1e-10f;
    // r = x<0 ? (x+|y|)/(|y|-x) : (x-|y|)/(x+|y|): select the operands, divide once (same IEEE op)
    const bool neg = xx < 0.0f;
    const float xpa = xx + ay;
    const float num = neg ? xpa : (xx - ay);
    const float den = neg ? (ay - xx) : xpa;
    const float rr = num / den;
    float ang = neg ? THRQTR_PI : ONEQTR_PI;
    ang = __builtin_fmaf(__builtin_fmaf(0.1963f * rr, rr, -0.9817f), rr, ang);
    const float nang = -1.0f * ang;
    ang = (b < 0.0f) ? nang : ang;
    ang = ang + ((ang < 0.0f) ? pi : 0.0f);
    const float sL1 = sqrt_approx<MODE>(L1, tab, P.lut, rare);
    const float sL2 = sqrt_approx<MODE>(L2, tab, P.lut, rare);
    const float coh = (sL1 - sL2) / ((sL1 + sL2) + 1e-17f);
    const float str = L1;
    const float fl = __builtin_floorf(ang * P.qangle);
    int ai = (fl >= -2147483648.0f && fl < 2147483648.0f) ? (int)fl : (int)0x80000000;   // cvtps_epi32
    ai = min(23, max(ai, 0));
    int si, ci;
    if (!LEGACY) {
        si = (int)(P.qs0 <= str) + (int)(P.qs1 <= str);
        ci = (int)(P.qc0 <= coh) + (int)(P.qc1 <= coh);
    } else {
        si = 2 - ((int)(str <= P.qs0) + (int)(str <= P.qs1));
        ci = 2 - ((int)(coh <= P.qc0) + (int)(coh <= P.qc1));
    }
    return ai * 9 + si * 3 + ci;
}

// out-of-line slow paths (kept out of the hot straight-line code)
__device__ __attribute__((noinline)) int hash_px_generic(float a, float b, float d, const HashQ P, const uint2* tab)
{
    bool unused = false;
    return hash_px_impl<2>(a, b, d, P, tab, unused);
}
__device__ __attribute__((noinline)) int hash_px_legacy(float a, float b, float d, const HashQ P, const uint2* tab)
{
    bool unused = false;
    return hash_px_impl<1>(a, b, d, P, tab, unused);
}

// ------------------------------------------------------------------------------------------------
// k_hash: structure tensor + hash.  Block = 256 threads = 4 waves; tile = 64 columns x 4R rows of
// the filtered zone [6,H-6) x [6,c_final); wave w owns rows [wR, wR+R), lane = column.
// Column accumulators S_k (k = patch column) are built sequentially over the 11 patch rows
// (computeGTWG_Segment_AVX512_32f, Raisr_AVX512.cpp:96-121) and folded in the association of
// sumitup_ps_512 (:37-44): sum = ((S1+S9)+S5 + (S7+S3)) + (((S0+S8)+S4) + ((S2+S10)+S6)).
// (The even/odd-pixel lane placements of the reference give the same value: they only commute
// operands of individual additions.)
// ------------------------------------------------------------------------------------------------
__constant__ int c_col_order[11] = {0, 8, 4, 2, 10, 6, 1, 9, 5, 7, 3};

// Gradient tile element (gx, gy).  (A binary16-packed tile -- 8-bit gradients are exact in binary16 and v_fma_mix_f32 gives
// the products directly -- halves the tile's LDS, but measured 3 % slower in k_hashfilter_ac and no faster in k_hash_ac.)
__device__ __forceinline__ f2 grad_load(const f2* p) { return *p; }
__device__ __forceinline__ void grad_store(f2* p, float gx, float gy) { *p = (f2){gx, gy}; }
__device__ __forceinline__ void grad_products(const f2* p, float& pa, float& pb, float& pd)
{
    const f2 g = *p;
    pa = g.x * g.x; pb = g.x * g.y; pd = g.y * g.y;
}
template <typename T> struct GradOf { using type = f2; };



// AVX2ALL: asm=avx2 frames -- every column takes the RCPPS/RSQRTPS flavour, inlined as straight-line code with both
// LUTs (8 KB) staged in LDS; otherwise the AVX-512 flavour with the out-of-line AVX2 replay of the tail columns.
//
// hash_phase: the work of one tile once its LR window (origin (r0-6, c0-6), row stride LW) is in sL.  Returns, per
// lane (= column c0+lane) and row j of the wave's R rows, hA = first hash (0xFF: pixel not filtered) and hB = the
// AVX2 re-hash of an overlap column (0xFF elsewhere).  Ends with every wave past its last LDS read of sG.
template <int R, bool AVX2ALL, int LW, typename GT = f2>
__device__ __forceinline__ void hash_phase(const PassParams& P, const GaussW& gw, const float* sL, GT* sG,
                                           const uint2* sTab, const uint16_t* sLut, int c0, int r0,
                                           unsigned (&hA)[R], unsigned (&hB)[R])
{
    constexpr int TH = 4 * R;
    constexpr int GW_ = 74, GH = TH + 10;   // gradient tile incl. 5-px halo
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // G(ty,tx) <-> image (r0-5+ty, c0-5+tx) <-> L tile (ty+1, tx+1)
    {
        auto grad = [&](int ty, int tx) {
            const float gxv = sL[(ty + 2) * LW + tx + 1] - sL[ty * LW + tx + 1];        // GetGx: row below - row above
            const float gyv = sL[(ty + 1) * LW + tx + 2] - sL[(ty + 1) * LW + tx];      // GetGy: right - left
            grad_store(&sG[ty * GW_ + tx], gxv, gyv);
        };
        const int wu = __builtin_amdgcn_readfirstlane(w);
        __builtin_assume(wu >= 0 && wu < 4);
#pragma unroll
        for (int it = 0; it < (GH + 3) / 4; it++)                                       // columns [0,64): wave = row
            if (wu + 4 * it < GH) grad(wu + 4 * it, lane);
        constexpr unsigned NR = GH * (GW_ - 64);
#pragma unroll
        for (unsigned it = 0; it < (NR + 255u) / 256u; it++) {                           // the 10 right-hand columns
            const unsigned idx = threadIdx.x + 256u * it;
            const int ty = (int)(idx / (GW_ - 64)), tx = 64 + (int)(idx - (unsigned)ty * (GW_ - 64));
            if (idx < NR) grad(ty, tx);
        }
    }
    __syncthreads();

    f2 curAD[R], holdAD[R], t1AD[R];
    float curB[R], holdB[R], t1B[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        curAD[j] = holdAD[j] = t1AD[j] = (f2){0.f, 0.f};
        curB[j] = holdB[j] = t1B[j] = 0.f;
    }
#pragma unroll 1
    for (int kk = 0; kk < 11; kk++) {
        const int k = c_col_order[kk];
        f2 g[R + 10];
#pragma unroll
        for (int t = 0; t < R + 10; t++) g[t] = grad_load(&sG[(w * R + t) * GW_ + lane + k]);
        f2 AD[R];
        float B[R];
#pragma unroll
        for (int j = 0; j < R; j++) { AD[j] = (f2){0.f, 0.f}; B[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float wv = gw.wT[k][i];
            const f2 w2 = {wv, wv};
#pragma unroll
            for (int j = 0; j < R; j++) {
                const f2 gg = g[i + j];
                const f2 pq = gg * w2;                                   // (gx*w, gy*w)
                AD[j] = __builtin_elementwise_fma(pq, gg, AD[j]);        // A += (gx*w)*gx ; D += (gy*w)*gy
                B[j] = __builtin_fmaf(pq.x, gg.y, B[j]);                 // B += (gx*w)*gy
            }
        }
        const bool start = (kk == 0) | (kk == 3) | (kk == 6) | (kk == 9);
        if (start) {
#pragma unroll
            for (int j = 0; j < R; j++) { curAD[j] = AD[j]; curB[j] = B[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) { curAD[j] = curAD[j] + AD[j]; curB[j] = curB[j] + B[j]; }
        }
        if (kk == 2 || kk == 8) {
#pragma unroll
            for (int j = 0; j < R; j++) { holdAD[j] = curAD[j]; holdB[j] = curB[j]; }
        }
        if (kk == 5) {
#pragma unroll
            for (int j = 0; j < R; j++) { t1AD[j] = holdAD[j] + curAD[j]; t1B[j] = holdB[j] + curB[j]; }
        }
    }

    const int c = c0 + lane;
    f2 ad[R];
    float bb[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        ad[j] = (holdAD[j] + curAD[j]) + t1AD[j];            // (Gb+Gc) + (Ga+Gd)
        bb[j] = (holdB[j] + curB[j]) + t1B[j];
    }
    const bool inB = c >= P.b_begin && c < P.b_end;
    if constexpr (AVX2ALL) {
        const HashQ HQ = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1, sLut};
#pragma unroll
        for (int j = 0; j < R; j++) {
            bool unused = false;
            const unsigned h = (unsigned)hash_px_impl<1>(ad[j].x, bb[j], ad[j].y, HQ, sTab, unused);
            const int r = r0 + w * R + j;
            hA[j] = (r < P.H - kMargin && c < P.c_final && inB) ? h : 0xFFu;
            hB[j] = 0xFFu;
        }
        return;
    }
    // AVX-512 flavour for every pixel of the lane as straight-line code (independent chains interleave);
    // the rare cases -- generic approximation-instruction inputs, AVX2 flavour of the tail columns -- follow.
    const HashQ HQ = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1, P.lut_legacy};
    const bool inA = c >= P.a_begin && c < P.a_end;
    unsigned h1[R];
    bool rare[R];
#pragma unroll
    for (int j = 0; j < R; j++) { h1[j] = 0xFFu; rare[j] = false; }
    if (P.a_end > P.a_begin) {                               // kernel-uniform
#pragma unroll
        for (int j = 0; j < R; j++) h1[j] = (unsigned)hash_px_impl<0>(ad[j].x, bb[j], ad[j].y, HQ, sTab, rare[j]);
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int r = r0 + w * R + j;
        hA[j] = 0xFFu; hB[j] = 0xFFu;
        if (r < P.H - kMargin && c < P.c_final) {
            unsigned h = h1[j];
            if (rare[j]) h = (unsigned)hash_px_generic(ad[j].x, bb[j], ad[j].y, HQ, sTab);
            if (inB) {                                       // the tail columns of AVX-512 mode
                const unsigned hL = (unsigned)hash_px_legacy(ad[j].x, bb[j], ad[j].y, HQ, sTab);
                if (inA) hB[j] = hL;                         // re-hashed tail column
                else h = hL;
            }
            hA[j] = (inA || inB) ? h : 0xFFu;
        }
    }
}

// stage the approximation tables of the hash flavour into LDS
template <bool AVX2ALL>
__device__ __forceinline__ void stage_hash_tables(const PassParams& P, uint2* sTab, uint16_t* sLut)
{
    if constexpr (AVX2ALL) {
        const uint2* src = reinterpret_cast<const uint2*>(P.lut_legacy);           // 4096 x u16 = 1024 x 8 B
        for (int i = threadIdx.x; i < 1024; i += 256) reinterpret_cast<uint2*>(sLut)[i] = src[i];
    } else {
        if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    }
}

template <int R, typename T, bool AVX2ALL>
__global__ __launch_bounds__(256, 4) void k_hash(const T* __restrict__ lr, PassParams P, GaussW gw,
                                                  uint8_t* __restrict__ hash_out, uint8_t* __restrict__ hash2_out)
{
    constexpr int TH = 4 * R;
    constexpr int LW = 76, LH = TH + 12;    // LR tile incl. 6-px halo
    __shared__ float sL[LH * LW];
    __shared__ f2 sG[(TH + 10) * 74];
    __shared__ uint2 sTab[AVX2ALL ? 1 : 128];
    __shared__ uint16_t sLut[AVX2ALL ? 4096 : 1];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * 64, r0 = kMargin + by * TH;

    stage_hash_tables<AVX2ALL>(P, sTab, sLut);
    stage_tile<LH, LW, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL);
    __syncthreads();
    unsigned hA[R], hB[R];
    hash_phase<R, AVX2ALL, LW>(P, gw, sL, sG, sTab, sLut, c0, r0, hA, hB);
    const int c = c0 + lane;
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int r = r0 + w * R + j;
        if (r < P.H - kMargin && c < P.c_final) {
            hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = (uint8_t)hA[j];
            if (hB[j] != 0xFFu) hash2_out[(size_t)r * 16 + (c - P.ov_begin)] = (uint8_t)hB[j];
        }
    }
}

// Test hook: the fp32 hash of arbitrary (a, b, d) tensor triples through exactly the code k_hash runs
// (fast path + generic fall-back for the AVX-512 flavour, or the AVX2 flavour).
__global__ __launch_bounds__(256) void k_debug_hash(const float* __restrict__ abd, unsigned n, PassParams P, int legacy,
                                                    uint8_t* __restrict__ out)
{
    __shared__ uint2 sTab[128];
    if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    __syncthreads();
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const HashQ HQ = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1, P.lut_legacy};
    const float a = abd[3 * (size_t)i], b = abd[3 * (size_t)i + 1], d = abd[3 * (size_t)i + 2];
    unsigned h;
    if (legacy) {
        h = (unsigned)hash_px_legacy(a, b, d, HQ, sTab);
    } else {
        bool rare = false;
        h = (unsigned)hash_px_impl<0>(a, b, d, HQ, sTab, rare);
        if (rare) h = (unsigned)hash_px_generic(a, b, d, HQ, sTab);
    }
    out[i] = (uint8_t)h;
}


// ------------------------------------------------------------------------------------------------
// Certified hashing ("approximate, then certify").  The output needs the exact BUCKET, not the exact
// tensor.  k_hashfilter_ac therefore computes the structure tensor cheaply -- the literal 11 x 11 table is a rank-1
// table up to its 6-digit truncation, so a separable 11 + 11-tap pass over the per-pixel gradient products
// (gx^2, gx gy, gy^2) reproduces (a, b, d) to a relative eps of a few 1e-6 -- evaluates the hash quantities in plain
// fp32 (native sqrt / rcp), and accepts the bucket only when rigorous bounds on
//     |quantity the reference's instruction sequence produces  -  quantity computed here|
// keep angle*24/pi, strength and coherence away from every bucket boundary and every discontinuity of the
// reference's code (sign of b, xx, L2, the "ang < 0" wrap).  Everything else -- a few per cent of the pixels on
// noisy content, 1-D structures and exactly symmetric patterns on synthetic content -- goes through a per-tile
// worklist to the exact code (exact_pixel below: the same operations in the same order as hash_phase).
// The bounds (tests/tools/certify_proto.py derives and validates them against the oracle):
//   tensor      |a-a'| <= eps a', |d-d'| <= eps d', |b-b'| <= eps T'/2,  T' = a'+d'
//               eps = 1.05 (eps_w + 48 u): eps_w = max |us_i us_k / w_ik - 1| over the ACTUAL fp32 constants of both paths,
//               16 u for the reference's 12-term fma chains + 4-level fold, 23 u for the separable passes (u = 2^-24).
//   root        reference: rad = fl(T^2/4 - (ad - b^2)) = R + eta, R = ((a-d)/2)^2 + b^2, |eta| <= 2 u T^2;  s = sqrt14(rad),
//               sqrt14(x) = sqrt(x)(1 + e), |e| <= E (1.0e-4 for VRCP14(VRSQRT14), 6.5e-4 for RCPPS(RSQRTPS); enumerated in
//               tests/test_certify_bounds.py).  With s* = sqrt(R'):   |s - s*| <= E_s = 1.42 eps T' + (2e-7 + eps^2) T'^2 / s* + 1.05 E s*
//   L1, L2, xx  |. - .*| <= E_L = E_s + (eps/2 + 6 u) T'   (L2* = (a'd' - b'^2) / L1* carries 4 u T' more)
//   coherence   t = sqrt(L2/L1): |coh - coh*| <= 2 t* (0.55 (E_L2/(L2*-E_L2) + E_L/(L1*-E_L)) + 2.4 E) + 2e-6, needs L2* > 2 E_L2
//   angle       rr = (xx - ay)/(xx + ay), |d rr| <= 2((ay+E_ay) E_L + (xx+E_L) E_ay) / (xx + ay - E_L - E_ay)^2 + 4 u,
//               |P'(rr)| < 1 for the cubic P  =>  |ang_raw - ang_raw*| <= d rr + 1.5e-6;  q = ang 24/pi: + 2e-5
// ------------------------------------------------------------------------------------------------
struct SepW {
    float us[11];                // separable weights, sqrt(NF) folded in: us[i] us[k] ~ wT[k][i]
    float es1, es2;              // 1.42 eps, 2e-7 + eps^2
    float eEL, eEb;              // eps/2 + 6 u, eps/2
    float e105[2], e24[2];       // 1.05 E and 2.4 E for the AVX-512 (0) and the AVX2 (1) approximation instructions
};

struct HashQf { float qangle, qs0, qs1, qc0, qc1; };

// returns true when the bucket is certified
__device__ __forceinline__ bool approx_hash(float a, float b, float d, const HashQf Q, const SepW& S, int fl, unsigned& bucket)
{
    const float U1 = 5.9604645e-8f;                      // 2^-24
    const float pi = 3.141592653f;
    const float ONEQTR_PI = (float)(3.14159265358979323846 / 4.0);
    const float T = a + d;
    const float m = 0.5f * (a - d);
    const float bb = b * b;
    const float R = __builtin_fmaf(m, m, bb);
    const float s = __builtin_amdgcn_sqrtf(R);
    const float hT = 0.5f * T;
    const float L1 = hT + s;
    const float rL1 = __builtin_amdgcn_rcpf(L1);
    const float det = __builtin_fmaf(a, d, -bb);
    const float L2 = det * rL1;                          // = T/2 - s without the cancellation
    const float rs = __builtin_amdgcn_rcpf(s);
    const float E_s = __builtin_fmaf(S.es1, T, __builtin_fmaf((S.es2 * T) * T, rs, S.e105[fl] * s));
    bool ok = (T > 0.0f) & (s > 0.0f) & (E_s <= 0.25f * s);
    const float E_L = __builtin_fmaf(S.eEL, T, E_s);
    const float E_L2 = __builtin_fmaf(4.0f * U1, T, E_L);
    // strength
    ok &= (__builtin_fabsf(L1 - Q.qs0) > E_L) & (__builtin_fabsf(L1 - Q.qs1) > E_L);
    const int si = (int)(Q.qs0 <= L1) + (int)(Q.qs1 <= L1);
    // coherence
    ok &= L2 > 2.0f * E_L2;
    const float t = __builtin_amdgcn_sqrtf(L2 * rL1);
    const float coh = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    const float dcoh = __builtin_fmaf(2.0f * t, __builtin_fmaf(0.55f, __builtin_fmaf(E_L2, __builtin_amdgcn_rcpf(L2 - E_L2), E_L * __builtin_amdgcn_rcpf(L1 - E_L)), S.e24[fl]), 2e-6f);
    ok &= (__builtin_fabsf(coh - Q.qc0) > dcoh) & (__builtin_fabsf(coh - Q.qc1) > dcoh);
    const int ci = (int)(Q.qc0 <= coh) + (int)(Q.qc1 <= coh);
    // angle
    const float E_b = S.eEb * T;
    const float ab = __builtin_fabsf(b);
    const float ay = ab + 1e-10f;
    const float xx = m >= 0.0f ? m + s : bb * __builtin_amdgcn_rcpf(s - m);        // (s - m)(s + m) = b^2
    const float E_ay = __builtin_fmaf(U1, ay, E_b);
    const float xpa = xx + ay;
    const float D = xpa - (E_L + E_ay);
    const float rr = (xx - ay) * __builtin_amdgcn_rcpf(xpa);
    const float rD = __builtin_amdgcn_rcpf(D);
    const float drr = __builtin_fmaf(2.0f * __builtin_fmaf(ay + E_ay, E_L, (xx + E_L) * E_ay), rD * rD, 4.0f * U1);
    const float ang_raw = __builtin_fmaf(__builtin_fmaf(0.1963f * rr, rr, -0.9817f), rr, ONEQTR_PI);
    const float dang = drr + 1.5e-6f;
    float ang = b < 0.0f ? -ang_raw : ang_raw;
    ang = ang < 0.0f ? ang + pi : ang;
    const float q = ang * Q.qangle;
    const float dq = __builtin_fmaf(Q.qangle, dang, 2e-5f);
    const float k = __builtin_fminf(__builtin_fmaxf(__builtin_floorf(q), 0.0f), 23.0f);
    const float fr = q - k;
    const bool c_ang = (xx > 2.0f * E_L) & (D > 0.0f) & (ab > E_b) & (__builtin_fabsf(ang_raw) > dang) &
                       ((k < 1.0f) | (fr > dq)) & ((k > 22.0f) | (1.0f - fr > dq));
    ok &= c_ang;                                         // (a window without any gx or gy has L2 = 0 and is never certified)
    bucket = (unsigned)((int)k * 9 + si * 3 + ci);
    return ok;
}

// first / second hash of a pixel in column c from its exact tensor: the flavour logic of hash_phase's epilogue
__device__ __forceinline__ void flavour_hash(const PassParams& P, const uint2* sTab, float a, float b, float d, int c, unsigned& hA, unsigned& hB)
{
    const HashQ HQ = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1, P.lut_legacy};
    const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
    hA = 0xFFu; hB = 0xFFu;
    unsigned h = 0xFFu;
    if (inA) {
        bool rare = false;
        h = (unsigned)hash_px_impl<0>(a, b, d, HQ, sTab, rare);
        if (rare) h = (unsigned)hash_px_generic(a, b, d, HQ, sTab);
    }
    if (inB) {
        const unsigned hL = (unsigned)hash_px_legacy(a, b, d, HQ, sTab);
        if (inA) hB = hL; else h = hL;
    }
    hA = (inA || inB) ? h : 0xFFu;
}

// Exact tensor of up to four worklist pixels per wave with 16 lanes per pixel: lane l < 11 of a group runs the
// reference's chain of patch column l (the 11 patch rows in order, weights wl[i] = wT[l][i]); the 11 column sums are
// folded in sumitup_ps_512's association with DPP row shifts (row_shl:n -- lane i reads lane i+n of its row of 16):
//   u = S + shl8(S): lanes 0..2 = S0+S8, S1+S9, S2+S10;  v = u + shl4(S): Ga, Gb, Gd;  lane 3: S3 + S7 = Gc;
//   r = z + shl2(z): lane 0 = Ga+Gd, lane 1 = Gb+Gc;  lane 0: (Ga+Gd) + (Gb+Gc)  [fp addition commutes bit for bit].
// The latency of one round is ~11 rows instead of 121 taps: what matters when only a handful of pixels per tile need it.
template <int CTRL>
__device__ __forceinline__ float row_shl(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float fold11(float S, bool lane3)
{
    const float u = S + row_shl<0x108>(S);
    const float v = u + row_shl<0x104>(S);
    const float wv = S + row_shl<0x104>(S);
    const float z = lane3 ? wv : v;
    const float r = z + row_shl<0x102>(z);
    return r + row_shl<0x101>(r);
}
template <typename GT>
__device__ __forceinline__ void exact_tensor16(const GT* sG, const float (&wl)[11], int prow, int pcol, int l, float& a, float& b, float& d)
{
    constexpr int GW_ = 74;
    const GT* base = sG + prow * GW_ + pcol + min(l, 10);
    f2 AD = {0.f, 0.f};
    float B = 0.f;
    f2 gg[11];
#pragma unroll
    for (int i = 0; i < 11; i++) gg[i] = grad_load(base + i * GW_);
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const f2 w2 = {wl[i], wl[i]};
        const f2 pq = gg[i] * w2;
        AD = __builtin_elementwise_fma(pq, gg[i], AD);
        B = __builtin_fmaf(pq.x, gg[i].y, B);
    }
    const bool lane3 = l == 3;
    a = fold11(AD.x, lane3);
    b = fold11(B, lane3);
    d = fold11(AD.y, lane3);
}

// Approximate structure tensor of the lane's 4 pixels (rows [4w, 4w+4) of the tile, column = lane): separable 11 + 11 taps
// on the gradient products.  V pass: lane-task (x, rg) = column x of the gradient tile, output rows [4 rg, 4 rg + 4),
// results as float4 per (channel, row group, column) in sV; workgroup barrier; H pass: lane = column, wave = row group.
template <typename GT>
__device__ __forceinline__ void tensor_ac(const SepW& S, const GT* sG, float4* sV, float (&ta)[4], float (&tb)[4], float (&td)[4])
{
    constexpr int GW_ = 74;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto vpass = [&](int x, int rg) {
        float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f}, vd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 14; t++) {
            float pa, pb, pd;
            grad_products(sG + (4 * rg + t) * GW_ + x, pa, pb, pd);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = t - r;
                if (i >= 0 && i < 11) {
                    va[r] = __builtin_fmaf(S.us[i], pa, va[r]);
                    vb[r] = __builtin_fmaf(S.us[i], pb, vb[r]);
                    vd[r] = __builtin_fmaf(S.us[i], pd, vd[r]);
                }
            }
        }
        sV[(0 * 4 + rg) * GW_ + x] = make_float4(va[0], va[1], va[2], va[3]);
        sV[(1 * 4 + rg) * GW_ + x] = make_float4(vb[0], vb[1], vb[2], vb[3]);
        sV[(2 * 4 + rg) * GW_ + x] = make_float4(vd[0], vd[1], vd[2], vd[3]);
    };
    vpass(lane, w);
    if (w == 0 && lane < 40) vpass(64 + lane % 10, lane / 10);       // the 10 halo columns of all four row groups
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) { ta[r] = 0.f; tb[r] = 0.f; td[r] = 0.f; }
    // software-pipelined by hand (next column's three float4 in flight during this column's 12 FMAs) and fenced per
    // column: left alone, the scheduler issues all 33 loads first and sinks the FMAs into the hash code -- 132 live VGPRs
    float4 xa = sV[(0 * 4 + w) * GW_ + lane], xb = sV[(1 * 4 + w) * GW_ + lane], xd = sV[(2 * 4 + w) * GW_ + lane];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        float4 na = xa, nb = xb, nd = xd;
        if (k < 10) {
            na = sV[(0 * 4 + w) * GW_ + lane + k + 1];
            nb = sV[(1 * 4 + w) * GW_ + lane + k + 1];
            nd = sV[(2 * 4 + w) * GW_ + lane + k + 1];
        }
        const float uk = S.us[k];
        ta[0] = __builtin_fmaf(uk, xa.x, ta[0]); ta[1] = __builtin_fmaf(uk, xa.y, ta[1]); ta[2] = __builtin_fmaf(uk, xa.z, ta[2]); ta[3] = __builtin_fmaf(uk, xa.w, ta[3]);
        tb[0] = __builtin_fmaf(uk, xb.x, tb[0]); tb[1] = __builtin_fmaf(uk, xb.y, tb[1]); tb[2] = __builtin_fmaf(uk, xb.z, tb[2]); tb[3] = __builtin_fmaf(uk, xb.w, tb[3]);
        td[0] = __builtin_fmaf(uk, xd.x, td[0]); td[1] = __builtin_fmaf(uk, xd.y, td[1]); td[2] = __builtin_fmaf(uk, xd.z, td[2]); td[3] = __builtin_fmaf(uk, xd.w, td[3]);
        xa = na; xb = nb; xd = nd;
        __builtin_amdgcn_sched_barrier(0);
    }
    // pin the 12 sums here: otherwise LLVM sinks the FMAs of rows 1..3 into the per-pixel hash code and keeps (spills) the loaded columns
#pragma unroll
    for (int r = 0; r < 4; r++) asm volatile("" : "+v"(ta[r]), "+v"(tb[r]), "+v"(td[r]));
}

// hash stage of k_hashfilter_ac for one tile: sG holds the gradient tile.  Leaves the buckets of the tile in sH / sH2
// (0xFF: not filtered / no re-hash) and ends with a workgroup barrier.
constexpr unsigned kListMax = 48;             // in-tile worklist of k_hashfilter_ac: more uncertain pixels -> the whole tile takes the exact routine
template <int LW, typename GT>
__device__ __forceinline__ void hash_phase_ac(const PassParams& P, const GaussW& gw, const SepW& S, const float* sL, GT* sG, float4* sV,
                                              const uint2* sTab, uint8_t* sH, uint8_t* sH2, uint16_t* sList, unsigned* sCnt,
                                              int c0, int r0)
{
    constexpr int GW_ = 74, TW = 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) sCnt[0] = 0;
    float ta[4], tb[4], td[4];
    tensor_ac(S, sG, sV, ta, tb, td);
    // ---- approximate hash + certification of the lane's 4 pixels ----
    const int c = c0 + lane;
    const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
    const int fl = inB ? 1 : 0;                            // the AVX2 flavour's wider table error covers the re-hashed columns too
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    unsigned nUnc = 0, certbits = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int prow = 4 * w + j;
        const int r = r0 + prow;
        const bool zone = r < P.H - kMargin && c < P.c_final && (inA || inB);
        unsigned bucket;
        bool cert = approx_hash(ta[j], tb[j], td[j], Q, S, fl, bucket);
        const bool zero = (ta[j] + td[j]) == 0.0f;          // flat window: the reference's tensor is exactly (0, 0, 0) as well
        cert |= zero;
        // first hash: AVX-512 flavour where the column has one, else the AVX2 flavour; second hash: AVX2 flavour of the
        // re-hashed columns.  A certified bucket holds for both flavours (fl selects the wider table error there); the
        // zero tensor's bucket is looked up per flavour (computed once per tile with the exact code).
        const unsigned bA = zero ? (unsigned)P.zero_bucket[inA ? 0 : 1] : bucket;
        const unsigned bB = zero ? (unsigned)P.zero_bucket[1] : bucket;
        const bool unc = zone && (!cert || P.cert_check);
        sH[prow * TW + lane] = zone ? (uint8_t)bA : (uint8_t)0xFFu;
        sH2[prow * TW + lane] = (zone && inA && inB) ? (uint8_t)bB : (uint8_t)0xFFu;
        certbits |= (cert ? 1u : 0u) << j;
        if (unc) {
            const unsigned slot = atomicAdd(sCnt, 1u);
            if (slot < kListMax) sList[slot] = (uint16_t)((prow << 6) | lane | (cert ? 0x8000 : 0));
        }
        nUnc += (zone && !cert) ? 1u : 0u;
    }
    if (P.cert_stats) {
        if (nUnc) atomicAdd(&sCnt[1], nUnc);
    }
    __syncthreads();

    // ---- worklist: the exact path for what could not be certified ----
    const unsigned n = sCnt[0];
    unsigned bad = 0;
    if (n <= kListMax) {
        // short list (the usual case): 16 lanes per pixel, four pixels per wave and round
        if (n) {
            const int g = lane >> 4, l = lane & 15;
            float wl[11];
#pragma unroll
            for (int i = 0; i < 11; i++) wl[i] = P.gauss_dev[min(l, 10) * 12 + i];
            for (unsigned rd = (unsigned)w; 4u * rd < n; rd += 4u) {
                const unsigned e = 4u * rd + (unsigned)g;
                const unsigned ent = sList[min(e, n - 1u)];
                const int prow = (ent >> 6) & 15, pcol = ent & 63;
                float a, b, d;
                exact_tensor16(sG, wl, prow, pcol, l, a, b, d);
                if (l == 0 && e < n) {
                    unsigned hA, hB;
                    flavour_hash(P, sTab, a, b, d, c0 + pcol, hA, hB);
                    if ((ent & 0x8000u) && (sH[prow * TW + pcol] != (uint8_t)hA || (hB != 0xFFu && sH2[prow * TW + pcol] != (uint8_t)hB))) bad++;
                    sH[prow * TW + pcol] = (uint8_t)hA;
                    sH2[prow * TW + pcol] = (uint8_t)hB;
                }
            }
        }
    } else {
        // long list (synthetic content, self-check mode): the whole tile through the all-exact routine (hash_phase; it
        // rebuilds the gradient tile from the LR window), every wave busy.  The AVX2 flavour takes its out-of-line path.
        unsigned hA[4], hB[4];
        hash_phase<4, false, LW, GT>(P, gw, sL, sG, sTab, nullptr, c0, r0, hA, hB);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int prow = 4 * w + j;
            if (hA[j] != 0xFFu && ((certbits >> j) & 1u) &&
                (sH[prow * TW + lane] != (uint8_t)hA[j] || (hB[j] != 0xFFu && sH2[prow * TW + lane] != (uint8_t)hB[j]))) bad++;
            sH[prow * TW + lane] = (uint8_t)hA[j];
            sH2[prow * TW + lane] = (uint8_t)hB[j];
        }
    }
    if (P.cert_stats && bad) atomicAdd(&sCnt[2], bad);
    if (n) __syncthreads();                                // (n is the same in every thread)
}

// Test hook: the certified hash stage's decision for arbitrary APPROXIMATE tensor triples (a', b', d'): bucket and whether
// it would be certified, by the very approx_hash the kernels run (flavour 0: AVX-512 table error, 1: AVX2).
__global__ __launch_bounds__(256) void k_debug_approx_hash(const float* __restrict__ abd, unsigned n, PassParams P, SepW S, int fl,
                                                           uint8_t* __restrict__ bucket_out, uint8_t* __restrict__ cert_out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    const float a = abd[3 * (size_t)i], b = abd[3 * (size_t)i + 1], d = abd[3 * (size_t)i + 2];
    unsigned bucket;
    bool cert = approx_hash(a, b, d, Q, S, fl, bucket);
    if ((a + d) == 0.0f) { cert = true; bucket = (unsigned)P.zero_bucket[fl]; }
    bucket_out[i] = (uint8_t)bucket;
    cert_out[i] = cert ? 1 : 0;
}

#include "raisr_fp16_kernels.h"

// ------------------------------------------------------------------------------------------------
// k_filter: HR = (lo < v < hi) ? v : LR with v = DotProdPatch(patch, bank[hash][type])
// (Raisr_AVX512.cpp:134-149; accept test Raisr.cpp:1196-1200).  16 lanes per pixel: lane l owns the
// reference's zmm lane l: acc = p[l]*f[l]; acc = fma(p[16c+l], f[16c+l], acc) for c=1..7;
// then sumitup_ps_512 as DPP row rotations by 8, 4, 2, 1.
// Tile = 64 columns x 16 rows; wave w owns tile rows [4w, 4w+4); each step handles 4 adjacent pixels.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float row_ror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v)      // DPP quad_perm: CTRL = a | b<<2 | c<<4 | d<<6, lane i of a quad reads lane CTRL_i
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float tree16(float acc)
{
    acc = acc + row_ror<0x128>(acc);    // row_ror:8  -> r8[i] = a[i] + a[i+8]
    acc = acc + row_ror<0x124>(acc);    // row_ror:4  -> r4[i] = r8[i] + r8[i+4]
    acc = acc + row_ror<0x122>(acc);    // row_ror:2  -> r2[i] = r4[i] + r4[i+2]
    acc = acc + row_ror<0x121>(acc);    // row_ror:1  -> r2[0] + r2[1]
    return acc;
}

// filter_phase: the work of one 64 x 16 tile once its LR window is in LDS -- sP points at window position
// (row r0-5, column c0-5), row stride LW -- and the tile's hashes are in sH / sH2 (0xFF = not filtered / no re-hash).
template <int LW>
__device__ __forceinline__ void filter_phase(const PassParams& P, const float* sL, const uint8_t* sH, const uint8_t* sH2,
                                             int c0, int r0, float* __restrict__ hr)
{
    constexpr int TW = 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, l = lane & 15;
    int off[8];
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
        const int k = 16 * ch + l;
        off[ch] = (k < kTaps) ? (k / 11) * LW + (k % 11) : 0;   // padding taps: coefficient is +0, any finite pixel will do
    }

    // 32-bit buffer addressing of the filter bank (one descriptor per wave, built from uniform values)
    const __amdgpu_buffer_rsrc_t bank_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(P.bank), 0, P.bank_bytes, 0x00020000);
    const int tcol = (P.pixel_types == 4) ? ((g + 1) & 1) : 0;        // (c-5)&1 with c = c0 + 4s + g, c0 even
    const unsigned lane_off = (unsigned)(tcol * kTapsPad + l) * 4u;   // byte offset of (type column part, zmm lane)
    const unsigned bank_stride = (unsigned)(P.pixel_types * kTapsPad * 4);   // bytes per hash bucket (<= 2048)

#pragma unroll 1
    for (int row = 0; row < 4; row++) {
        const int prow = 4 * w + row;
        const int r = r0 + prow;
        const unsigned trow_off = (P.pixel_types == 4) ? (unsigned)(((r - 5) & 1) * 2 * kTapsPad * 4) : 0u;
        const unsigned row_lane_off = trow_off + lane_off;
        // LDS byte addresses of this lane's 8 taps (and the centre pixel) for step 0; step s adds the immediate 16*s
        const char* tap[8];
#pragma unroll
        for (int ch = 0; ch < 8; ch++) tap[ch] = reinterpret_cast<const char*>(sL + prow * LW + g + off[ch]);
        const char* ctr = reinterpret_cast<const char*>(sL + prow * LW + g + 5 * LW + 5);
#define RAISR_LDS_F(p, s) (*reinterpret_cast<const float*>((p) + 16 * (s)))
#define RAISR_BANK_F(voff) __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bank_rsrc, (voff), 0, 0))
        float keep = 0.0f;
        const bool anyB = sH2[prow * TW + lane] != 0xFFu;      // does this tile row contain re-hashed (tail) columns?
        // The 16 steps (4 adjacent pixels each) go in four groups {j, j+4, j+8, j+12}: the lane that keeps step s is
        // l == s, so the four steps of a group end in four different quads of the pixel's 16 lanes.  Each step's
        // accumulator is folded by the first two tree levels (row_ror 8, 4: every lane then holds r4[l & 3]), the four
        // steps are merged quad-wise into one register (quad m <- step j+4m), and the last two levels, the accept
        // test and the keep-select run once per group instead of once per step.  Same additions, same order.
        const char* ctrq = ctr + 64 * (l >> 2);                // centre pixel of the step this lane's quad ends up with
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float part[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int s = j + 4 * m;
                const unsigned hA = sH[prow * TW + 4 * s + g];
                // No branch for hA == 0xFF (pixel not filtered): its offset lies past the bank, the bounds-checked buffer
                // loads return +0, v = 0 fails the accept test (clamp_lo >= 0, checked at configure) and the pixel keeps LR.
                const unsigned voff = __umul24(hA, bank_stride) + row_lane_off;       // v_mad_u32_u24 (the 32x32 form is a slow 64-bit mad)
                float acc = RAISR_LDS_F(tap[0], s) * RAISR_BANK_F(voff);
#pragma unroll
                for (int ch = 1; ch < 8; ch++) acc = __builtin_fmaf(RAISR_LDS_F(tap[ch], s), RAISR_BANK_F(voff + 64u * ch), acc);
                acc = acc + row_ror<0x128>(acc);               // r8[i] = a[i] + a[i+8]
                part[m] = acc + row_ror<0x124>(acc);           // r4[i] = r8[i] + r8[i+4]   (period 4 over the 16 lanes)
            }
            float v = part[0];
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(part[1]), "s"(0x00f000f000f000f0ull));
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(part[2]), "s"(0x0f000f000f000f00ull));
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(part[3]), "s"(0xf000f000f000f000ull));
            v = v + quad_perm<0x4e>(v);                        // [2,3,0,1]: r2 = r4[i] + r4[i+2]
            v = v + quad_perm<0xb1>(v);                        // [1,0,3,2]: r2[0] + r2[1]
            float res = RAISR_LDS_F(ctrq, j);
            if (v > P.lo && v < P.hi) res = v;
            // lane (g,l) keeps pixel column 4l+g, i.e. step l: in group j those are the lanes with (l & 3) == j
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(keep) : "v"(res), "s"(0x1111111111111111ull << j));
        }
        if (__any(anyB)) {                                      // tail columns only: AVX2 re-hash (keep-first-if-rejected;
#pragma unroll 1                                                 //  Randomness blends the last candidate instead)
            for (int s = 0; s < 16; s++) {
                const unsigned hB = sH2[prow * TW + 4 * s + g];
                if (hB == 0xFFu) continue;
                const unsigned voff = __umul24(hB, bank_stride) + row_lane_off;
                float acc = RAISR_LDS_F(tap[0], s) * RAISR_BANK_F(voff);
#pragma unroll
                for (int ch = 1; ch < 8; ch++) acc = __builtin_fmaf(RAISR_LDS_F(tap[ch], s), RAISR_BANK_F(voff + 64u * ch), acc);
                const float v = tree16(acc);
                if (s == l) {
                    if (v > P.lo && v < P.hi) keep = v;
                    else if (P.randomness) keep = RAISR_LDS_F(ctr, s);
                }
            }
        }
#undef RAISR_LDS_F
#undef RAISR_BANK_F
        const int c = c0 + 4 * l + g;
        if (r < P.H - kMargin && c < P.c_final) hr[(size_t)r * P.hr_pitch + c] = keep;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_filter(const T* __restrict__ lr, const uint8_t* __restrict__ hash,
                                                PassParams P, float* __restrict__ hr, unsigned* __restrict__ fix_counters = nullptr)
{
    // split pipeline: this launch follows the fix kernels in stream order, so their list counters can be cleared here
    if (fix_counters && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) fix_counters[0] = 0;
    constexpr int TW = 64, TH = 16, LW = TW + 11, LH = TH + 10;   // odd stride: fewer LDS bank conflicts on the patch reads
    __shared__ float sL[LH * LW];
    __shared__ uint8_t sH[TH * TW];         // first hash (0xFF = not filtered)
    __shared__ uint8_t sH2[TH * TW];        // second hash of the overlap columns (0xFF elsewhere)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;

    stage_tile<LH, TW + 10, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 5, c0 - 5, sL);
    for (int ty = w; ty < TH; ty += 4) {
        const int r = r0 + ty, c = c0 + lane;
        const bool in = r < P.H - kMargin && c < P.c_final;
        sH[ty * TW + lane] = in ? hash[(size_t)r * P.hash_pitch + c] : (uint8_t)0xFFu;
        sH2[ty * TW + lane] = (in && c >= P.ov_begin && c < P.ov_end) ? P.hash2[(size_t)r * 16 + (c - P.ov_begin)] : (uint8_t)0xFFu;
    }
    __syncthreads();
    filter_phase<LW>(P, sL, sH, sH2, c0, r0, hr);
}

// k_hashfilter: both stages of a 64 x 16 tile in one kernel.  The tensor/hash stage is fp32-VALU bound and the
// filter stage vector-L1 bound; with workgroups of one kernel in different stages on the same CU the two
// resources are busy at the same time, which separate launches only achieve by accident across streams.
// One LR window (6-px halo, stride 77: odd for the filter's patch reads) serves both stages; the hashes go from
// registers to LDS, and to the hash plane only when a test asks for it.
template <typename T, bool AVX2ALL>
__global__ __launch_bounds__(256, 4) void k_hashfilter(const T* __restrict__ lr, PassParams P, GaussW gw,
                                                       uint8_t* __restrict__ hash_out, float* __restrict__ hr)
{
    constexpr int R = 4, TW = 64, TH = 16;
    constexpr int LW = 77, LH = TH + 12;
    __shared__ float sL[LH * LW];
    __shared__ f2 sG[(TH + 10) * 74];
    __shared__ uint2 sTab[AVX2ALL ? 1 : 128];
    __shared__ uint16_t sLut[AVX2ALL ? 4096 : 1];
    __shared__ uint8_t sH[TH * TW];          // first hash; rows [4w, 4w+4) are written AND read by wave w only
    __shared__ uint8_t sH2[TH * TW];         // AVX2 re-hash of the overlap columns

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;

    stage_hash_tables<AVX2ALL>(P, sTab, sLut);
    stage_tile<LH, 76, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL);
    __syncthreads();
    unsigned hA[R], hB[R];
    hash_phase<R, AVX2ALL, LW>(P, gw, sL, sG, sTab, sLut, c0, r0, hA, hB);
    // A wave filters exactly the rows it hashed (rows [4w, 4w+4)), so no workgroup barrier separates the stages: the
    // four waves of a tile drift apart and the VALU-bound and the L1-bound stage overlap inside the workgroup too.
    const int c = c0 + lane;
#pragma unroll
    for (int j = 0; j < R; j++) {
        sH[(w * R + j) * TW + lane] = (uint8_t)hA[j];
        sH2[(w * R + j) * TW + lane] = (uint8_t)hB[j];
        const int r = r0 + w * R + j;
        if (P.write_hash && r < P.H - kMargin && c < P.c_final) hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = (uint8_t)hA[j];
    }
    __builtin_amdgcn_wave_barrier();                         // LDS is in order within a wave; keep the compiler from reordering
    filter_phase<LW>(P, sL + LW + 1, sH, sH2, c0, r0, hr);
}


// k_hashfilter_ac: k_hashfilter with the certified hash stage (hash_phase_ac) -- the production kernel of the fp32
// numerics.  Same tile, same LR window, same filter stage; the structure tensor costs ~90 instead of ~605 lane-ops per
// pixel and the hash ~100 instead of ~200; the few pixels whose bucket cannot be certified take the exact code.
// PART (profiling aid, RAISR_HIP_AC_PART): 0 = the production kernel, 1 = hash stage only, 2 = filter stage only (bucket 0).
template <typename T, int PART = 0>
__global__ __launch_bounds__(256, 3) void k_hashfilter_ac(const T* __restrict__ lr, PassParams P, GaussW gw, SepW S,
                                                          uint8_t* __restrict__ hash_out, float* __restrict__ hr)
{
    constexpr int TW = 64, TH = 16;
    constexpr int LW = 77, LH = TH + 12, GW_ = 74, GH = TH + 10;
    __shared__ float sL[LH * LW];
    using GT = typename GradOf<T>::type;
    __shared__ GT sG[GH * GW_];
    __shared__ float4 sV[3 * 4 * GW_];
    __shared__ uint2 sTab[128];
    __shared__ uint8_t sH[TH * TW];
    __shared__ uint8_t sH2[TH * TW];
    __shared__ uint16_t sList[kListMax];      // worklist entries
    __shared__ unsigned sCnt[3];              // worklist length; uncertain pixels; certified-but-wrong (check mode)

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;

    if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    if (threadIdx.x < 3) sCnt[threadIdx.x] = 0;
    stage_tile<LH, 76, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL);
    __syncthreads();
    {   // gradient tile: G(ty,tx) <-> image (r0-5+ty, c0-5+tx) <-> L tile (ty+1, tx+1)
        auto grad = [&](int ty, int tx) {
            const float gxv = sL[(ty + 2) * LW + tx + 1] - sL[ty * LW + tx + 1];
            const float gyv = sL[(ty + 1) * LW + tx + 2] - sL[(ty + 1) * LW + tx];
            grad_store(&sG[ty * GW_ + tx], gxv, gyv);
        };
        const int wu = __builtin_amdgcn_readfirstlane(w);
        __builtin_assume(wu >= 0 && wu < 4);
#pragma unroll
        for (int it = 0; it < (GH + 3) / 4; it++)
            if (wu + 4 * it < GH) grad(wu + 4 * it, lane);
        constexpr unsigned NR = GH * (GW_ - 64);
#pragma unroll
        for (unsigned it = 0; it < (NR + 255u) / 256u; it++) {
            const unsigned idx = threadIdx.x + 256u * it;
            const int ty = (int)(idx / (GW_ - 64)), tx = 64 + (int)(idx - (unsigned)ty * (GW_ - 64));
            if (idx < NR) grad(ty, tx);
        }
    }
    __syncthreads();
    if (PART != 2) hash_phase_ac<LW, GT>(P, gw, S, sL, sG, sV, sTab, sH, sH2, sList, sCnt, c0, r0);
    else {
        for (int i = threadIdx.x; i < TH * TW; i += 256) { sH[i] = (uint8_t)((i * 7) % 216); sH2[i] = 0xFFu; }
        __syncthreads();
    }
    if (P.write_hash) {
        const int c = c0 + lane;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + 4 * w + j;
            if (r < P.H - kMargin && c < P.c_final) hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = sH[(4 * w + j) * TW + lane];
        }
    }
    if (P.cert_stats && threadIdx.x == 0) {
        const int zr = min(TH, P.H - kMargin - r0), zc = min(TW, P.c_final - c0);
        if (sCnt[1]) atomicAdd(&P.cert_stats[0], sCnt[1]);
        if (sCnt[2]) atomicAdd(&P.cert_stats[1], sCnt[2]);
        atomicAdd(&P.cert_stats[2], (unsigned)(max(zr, 0) * max(zc, 0)));
    }
    if (PART != 1) filter_phase<LW>(P, sL + LW + 1, sH, sH2, c0, r0, hr);
    else if (sH[threadIdx.x] == 0xFEu) hr[0] = 0.f;       // keep the hash stage alive
}


// ------------------------------------------------------------------------------------------------
// Split pipeline of the fp32 numerics:  k_hash_ac -> k_fix_sparse -> k_fix_dense -> filter kernel.
// k_hash_ac computes the approximate tensor and the certified buckets of a 64 x 16 tile and writes one bucket per pixel
// (HBM, 1 B/pixel).  Pixels it cannot certify do NOT stall the tile: a tile with few of them appends their coordinates
// to a frame-level list (k_fix_sparse: exact tensor with 16 lanes per pixel, straight from the LR plane), a tile with
// many appends itself to the tile list (k_fix_dense: the all-exact hash_phase on those tiles).  Both lists are usually
// short or empty; the fix kernels are persistent grids that read the counts on the device.
// ------------------------------------------------------------------------------------------------
struct FixLists {
    unsigned* counts;            // per tile: number of listed pixels (0 .. kSparseMax), or kDenseTile; written by k_hash_ac every frame
    unsigned* sparse;            // [tile][kSparseMax]: (row << 16) | column
    unsigned* dense;             // (tile row << 16) | tile column
    unsigned* counters;          // [0] number of dense tiles; zeroed by the filter kernel that follows
    uint8_t* cert_mask;          // self-check mode: 1 where the bucket in the plane was certified (else null)
    int tiles_x, tiles_y;
};
constexpr unsigned kSparseMax = 96;          // a tile with more uncertain pixels than this is re-hashed as a whole
constexpr unsigned kDenseTile = 0xFFFFFFFFu;

// xcd_tile for a linear tile index t of a persistent grid whose size is a multiple of 8 (so t % 8 == blockIdx.x % 8)
__device__ __forceinline__ void xcd_tile_of(unsigned t, unsigned gx, unsigned n, int& bx, int& by)
{
    const unsigned n8 = n & ~7u;
    const unsigned u = t < n8 ? (t & 7u) * (n8 >> 3) + (t >> 3) : t;
    by = (int)(u / gx);
    bx = (int)(u - (unsigned)by * gx);
}

// Persistent workgroups: each walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and fetches the next tile's LR window
// into registers while it computes the current one.
template <typename T>
__global__ __launch_bounds__(256, 5) void k_hash_ac(const T* __restrict__ lr, PassParams P, SepW S, FixLists F,
                                                    uint8_t* __restrict__ hash_out, uint8_t* __restrict__ hash2_out)
{
    constexpr int TW = 64, TH = 16;
    constexpr int LW = 77, LH = TH + 12, GW_ = 74, GH = TH + 10;
    __shared__ float sL[LH * LW];             // 8624 B; after the gradient stage: worklist [1024 x u16]
    using GT = typename GradOf<T>::type;
    __shared__ GT sG[GH * GW_];
    __shared__ float4 sV[3 * 4 * GW_];
    __shared__ unsigned sCnt[2];
    uint16_t* sList = reinterpret_cast<uint16_t*>(sL);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned ntiles = (unsigned)(F.tiles_x * F.tiles_y);
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    TileRegs<LH, 76, T> R;
    unsigned t = blockIdx.x;
    int bx = 0, by = 0;
    if (t < ntiles) {
        xcd_tile_of(t, (unsigned)F.tiles_x, ntiles, bx, by);
        load_tile<LH, 76>(lr, P.lr_pitch, P.W, P.H, by * TH, bx * TW, R);      // window origin (r0 - 6, c0 - 6) = (by TH, bx TW)
    }
    for (; t < ntiles; t += gridDim.x) {
        const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;
        const unsigned tile_id = (unsigned)by * (unsigned)F.tiles_x + (unsigned)bx;
        const unsigned tile_pos = ((unsigned)by << 16) | (unsigned)bx;
        if (threadIdx.x < 2) sCnt[threadIdx.x] = 0;
        store_tile<LH, 76, LW>(R, sL);
        lds_barrier();
        if (t + gridDim.x < ntiles) {                       // next tile's window: in flight during this tile's arithmetic
            xcd_tile_of(t + gridDim.x, (unsigned)F.tiles_x, ntiles, bx, by);
            load_tile<LH, 76>(lr, P.lr_pitch, P.W, P.H, by * TH, bx * TW, R);
        }
        {   // gradient tile: G(ty,tx) <-> image (r0-5+ty, c0-5+tx) <-> L tile (ty+1, tx+1)
            auto grad = [&](int ty, int tx) {
                const float gxv = sL[(ty + 2) * LW + tx + 1] - sL[ty * LW + tx + 1];
                const float gyv = sL[(ty + 1) * LW + tx + 2] - sL[(ty + 1) * LW + tx];
                grad_store(&sG[ty * GW_ + tx], gxv, gyv);
            };
            const int wu = __builtin_amdgcn_readfirstlane(w);
            __builtin_assume(wu >= 0 && wu < 4);
#pragma unroll
            for (int it = 0; it < (GH + 3) / 4; it++)
                if (wu + 4 * it < GH) grad(wu + 4 * it, lane);
            constexpr unsigned NR = GH * (GW_ - 64);
#pragma unroll
            for (unsigned it = 0; it < (NR + 255u) / 256u; it++) {
                const unsigned idx = threadIdx.x + 256u * it;
                const int ty = (int)(idx / (GW_ - 64)), tx = 64 + (int)(idx - (unsigned)ty * (GW_ - 64));
                if (idx < NR) grad(ty, tx);
            }
        }
        lds_barrier();
        float ta[4], tb[4], td[4];
        tensor_ac(S, sG, sV, ta, tb, td);

        const int c = c0 + lane;
        const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
        const int fl = inB ? 1 : 0;
        unsigned nUnc = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int prow = 4 * w + j;
            const int r = r0 + prow;
            const bool zone = r < P.H - kMargin && c < P.c_final && (inA || inB);
            unsigned bucket;
            bool cert = approx_hash(ta[j], tb[j], td[j], Q, S, fl, bucket);
            const bool zero = (ta[j] + td[j]) == 0.0f;
            cert |= zero;
            const unsigned bA = zero ? (unsigned)P.zero_bucket[inA ? 0 : 1] : bucket;
            const unsigned bB = zero ? (unsigned)P.zero_bucket[1] : bucket;
            if (r < P.H - kMargin && c < P.c_final) {
                hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = zone ? (uint8_t)bA : (uint8_t)0xFFu;
                if (inA && inB) hash2_out[(size_t)r * 16 + (c - P.ov_begin)] = (uint8_t)bB;
                if (F.cert_mask) F.cert_mask[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = (uint8_t)(cert ? 1 : 0);
            }
            if (zone && (!cert || P.cert_check)) {
                const unsigned slot = atomicAdd(&sCnt[0], 1u);
                sList[slot] = (uint16_t)((prow << 6) | lane);
            }
            nUnc += (zone && !cert) ? 1u : 0u;
        }
        if (P.cert_stats && nUnc) atomicAdd(&sCnt[1], nUnc);
        lds_barrier();
        const unsigned n = sCnt[0];
        if (n <= kSparseMax) {
            if (threadIdx.x < n) {
                const unsigned ent = sList[threadIdx.x];
                F.sparse[tile_id * kSparseMax + threadIdx.x] = ((unsigned)(r0 + (int)((ent >> 6) & 15)) << 16) | (unsigned)(c0 + (int)(ent & 63));
            }
            if (threadIdx.x == 0) F.counts[tile_id] = n;
        } else if (threadIdx.x == 0) {
            F.counts[tile_id] = kDenseTile;
            F.dense[atomicAdd(&F.counters[0], 1u)] = tile_pos;
        }
        if (P.cert_stats && threadIdx.x == 0) {
            const int zr = min(TH, P.H - kMargin - r0), zc = min(TW, P.c_final - c0);
            if (sCnt[1]) atomicAdd(&P.cert_stats[0], sCnt[1]);
            atomicAdd(&P.cert_stats[2], (unsigned)(max(zr, 0) * max(zc, 0)));
        }
        lds_barrier();                                     // worklist (in the LR window's space) and counters are free again
    }
}

// k_fix_sparse: the reference's exact tensor + hash for the listed pixels of one tile per wave: tensors with 16 lanes per
// pixel (exact_tensor16's scheme, the 13 x 13 LR window read straight from the L2-resident plane), parked in LDS, then
// one hash pass with a lane per pixel.
template <typename T>
__global__ __launch_bounds__(256) void k_fix_sparse(const T* __restrict__ lr, PassParams P, FixLists F,
                                                    uint8_t* __restrict__ hash_out, uint8_t* __restrict__ hash2_out)
{
    __shared__ uint2 sTab[128];
    __shared__ float sABD[4][kSparseMax][3];
    if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, l = lane & 15, lc = min(l, 10);
    const unsigned tile = blockIdx.x * 4u + (unsigned)wv;
    if (tile >= (unsigned)(F.tiles_x * F.tiles_y)) return;
    const unsigned n = F.counts[tile];
    if (n == 0 || n == kDenseTile) return;
    const unsigned* list = F.sparse + (size_t)tile * kSparseMax;
    float wl[11];
#pragma unroll
    for (int i = 0; i < 11; i++) wl[i] = P.gauss_dev[lc * 12 + i];
    for (unsigned rd = 0; 4u * rd < n; rd++) {
        const unsigned e = 4u * rd + (unsigned)g;
        const unsigned ent = list[min(e, n - 1u)];
        const int r = (int)(ent >> 16), c = (int)(ent & 0xFFFFu);
        // column x = c - 5 + l of the window: rows r-6 .. r+6 of it, rows r-5 .. r+5 of its two neighbours
        const T* col = lr + (unsigned)(r - 6) * (unsigned)P.lr_pitch + (unsigned)(c - 5 + lc);
        float Lc[13], Ll[11], Lr[11];
#pragma unroll
        for (int j = 0; j < 13; j++) Lc[j] = (float)col[(unsigned)j * (unsigned)P.lr_pitch];
#pragma unroll
        for (int i = 0; i < 11; i++) {
            Ll[i] = (float)col[(unsigned)(i + 1) * (unsigned)P.lr_pitch - 1];
            Lr[i] = (float)col[(unsigned)(i + 1) * (unsigned)P.lr_pitch + 1];
        }
        f2 AD = {0.f, 0.f};
        float B = 0.f;
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const f2 gg = {Lc[i + 2] - Lc[i], Lr[i] - Ll[i]};          // GetGx: row below - row above; GetGy: right - left
            const f2 w2 = {wl[i], wl[i]};
            const f2 pq = gg * w2;
            AD = __builtin_elementwise_fma(pq, gg, AD);
            B = __builtin_fmaf(pq.x, gg.y, B);
        }
        const bool lane3 = l == 3;
        const float a = fold11(AD.x, lane3), b = fold11(B, lane3), d = fold11(AD.y, lane3);
        if (l == 0 && e < n) { sABD[wv][e][0] = a; sABD[wv][e][1] = b; sABD[wv][e][2] = d; }
    }
    __builtin_amdgcn_wave_barrier();                          // LDS is in order within a wave
    unsigned bad = 0;
    for (unsigned e = (unsigned)lane; e < n; e += 64u) {
        const unsigned ent = list[e];
        const int r = (int)(ent >> 16), c = (int)(ent & 0xFFFFu);
        unsigned hA, hB;
        flavour_hash(P, sTab, sABD[wv][e][0], sABD[wv][e][1], sABD[wv][e][2], c, hA, hB);
        const unsigned idx = (unsigned)r * (unsigned)P.hash_pitch + (unsigned)c;
        if (F.cert_mask && F.cert_mask[idx] && (hash_out[idx] != (uint8_t)hA || (hB != 0xFFu && hash2_out[(size_t)r * 16 + (c - P.ov_begin)] != (uint8_t)hB))) bad++;
        hash_out[idx] = (uint8_t)hA;
        if (hB != 0xFFu) hash2_out[(size_t)r * 16 + (c - P.ov_begin)] = (uint8_t)hB;
    }
    if (P.cert_stats && bad) atomicAdd(&P.cert_stats[1], bad);
}

// k_fix_dense: the all-exact hash stage (hash_phase) for the listed tiles.  Persistent grid.
template <typename T, bool AVX2ALL>
__global__ __launch_bounds__(256, 4) void k_fix_dense(const T* __restrict__ lr, PassParams P, GaussW gw, FixLists F,
                                                      uint8_t* __restrict__ hash_out, uint8_t* __restrict__ hash2_out)
{
    constexpr int R = 4, TH = 4 * R;
    constexpr int LW = 76, LH = TH + 12;
    __shared__ float sL[LH * LW];
    __shared__ f2 sG[(TH + 10) * 74];
    __shared__ uint2 sTab[AVX2ALL ? 1 : 128];
    __shared__ uint16_t sLut[AVX2ALL ? 4096 : 1];
    const unsigned n = F.counters[0];
    if (blockIdx.x >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    stage_hash_tables<AVX2ALL>(P, sTab, sLut);
    unsigned bad = 0;
    for (unsigned t = blockIdx.x; t < n; t += gridDim.x) {
        const unsigned tile = F.dense[t];
        const int c0 = kMargin + (int)(tile & 0xFFFFu) * 64, r0 = kMargin + (int)(tile >> 16) * TH;
        __syncthreads();                                    // the previous tile's LDS reads are done
        stage_tile<LH, LW, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL);
        __syncthreads();
        unsigned hA[R], hB[R];
        hash_phase<R, AVX2ALL, LW>(P, gw, sL, sG, sTab, sLut, c0, r0, hA, hB);
        const int c = c0 + lane;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int r = r0 + w * R + j;
            if (r < P.H - kMargin && c < P.c_final) {
                const unsigned idx = (unsigned)r * (unsigned)P.hash_pitch + (unsigned)c;
                if (F.cert_mask && F.cert_mask[idx] && hA[j] != 0xFFu &&
                    (hash_out[idx] != (uint8_t)hA[j] || (hB[j] != 0xFFu && hash2_out[(size_t)r * 16 + (c - P.ov_begin)] != (uint8_t)hB[j]))) bad++;
                hash_out[idx] = (uint8_t)hA[j];
                if (hB[j] != 0xFFu) hash2_out[(size_t)r * 16 + (c - P.ov_begin)] = (uint8_t)hB[j];
            }
        }
    }
    if (P.cert_stats && bad) atomicAdd(&P.cert_stats[1], bad);
}


// ------------------------------------------------------------------------------------------------
// k_filter_lds16: the filter stage with the filter bank in LDS (north_star: "per-CU LDS cache"), split pipeline only.
// The stand-alone k_filter is bound by the vector L1: 512 B of coefficients per pixel at 64 B/clk/CU.  One pixel
// type's bank is 216 x 128 floats = 108 KB and fits the 160 KB LDS, where a lane fetches its 8 coefficients with two
// ds_read_b128 (256 B/clk/CU).  So: persistent workgroups of 16 waves, one per CU, each owning ONE pixel type
// (blockIdx & 3): it loads that type's bank once per launch and walks the tiles of its type -- 64 x 16 pixels of the
// type = a 128 x 32 pixel region of the plane (SP = 2; ratio 1.5 has a single type and SP = 1) -- with the LR window of
// the next tile (and its buckets) prefetched into registers while the current one is filtered.  Wave q of the workgroup
// filters row q of the tile.  LDS: bank 217 rows (row 216 = zeros: "not filtered") 111 104 B + 2 x LR window + 2 x bucket tiles.
// The LR window is held as binary16 with TWO pixels per lane and step:
// 8- and 10-bit samples are exact in binary16, and v_fma_mix_f32 multiplies a binary16 operand (either half of a
// VGPR) into an fp32 FMA -- bit for bit the fp32 FMA of the converted value.  The window is stored de-interleaved by
// column parity (pixels of one type are SP columns apart, so the two pixels a lane works on are neighbours in their
// parity plane), each plane twice: as is, and shifted by one sample, so that every (even-aligned) 4-byte read returns
// the pair a lane needs.  Per 8 pixels: 8 ds_read_b32 (patch pairs) + 4 ds_read_b128 (coefficients) = 32 LDS cycles
// instead of 48.  Pixel m = 8 ds + 2 g + e (ds = step 0..7, g = lane group, e = half); lane (g, l) keeps m with l = 2 ds + e.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_mix_lo(unsigned pair, float f, float acc)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(f), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned pair, float f, float acc)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(f), "v"(acc));
    return d;
}

template <typename T, int SP>
__global__ __launch_bounds__(1024) void k_filter_lds16(const T* __restrict__ lr, const uint8_t* __restrict__ hash, PassParams P,
                                                       float* __restrict__ hr, unsigned* __restrict__ fix_counters)
{
    constexpr int TW = 64, TH = 16;                                  // pixels of the type per tile
    constexpr int WW = SP * (TW - 1) + 11, WH = SP * (TH - 1) + 11;  // LR window of a tile: 137 x 41 (SP = 2), 74 x 26 (SP = 1)
    // Row layout in binary16 samples: one region per (plane, copy), 70 (SP = 2) / 74 (SP = 1) samples each, placed so that the 32
    // dwords a half-wave's ds_read_b32 touches (16 taps x 2 lane groups) fall into 32 different banks for every chunk
    // (exhaustive search over region orders, gaps and row strides; SQ_LDS_BANK_CONFLICT 33 % -> ~0 of the LDS cycles)
    constexpr int RS = SP == 2 ? 286 : 154;
    constexpr int REG0 = SP == 2 ? 142 : 0, REG1 = SP == 2 ? 214 : 78, REG2 = 0, REG3 = 72;    // region of (plane * 2 + copy)
    constexpr int NLOAD = (WW * WH + 1023) / 1024;
    extern __shared__ float smem[];
    float* sBank = smem;                                             // [217][2][16][4]
    uint16_t* sT0 = reinterpret_cast<uint16_t*>(sBank + 217 * 128);
    uint16_t* sT1 = sT0 + WH * RS;
    uint8_t* sHb = reinterpret_cast<uint8_t*>(sT1 + WH * RS);        // [2][2][TH * TW]: buffer, {first, second hash}

    if (fix_counters && blockIdx.x == 0 && threadIdx.x == 0) fix_counters[0] = 0;   // follows the fix kernels in stream order
    const int ntypes = SP * SP;
    const int type = (int)(blockIdx.x % (unsigned)ntypes);
    const int tr = type >> 1, tc = type & 1;
    const int rbase = SP == 2 ? kMargin + (tr ^ 1) : kMargin;
    const int cbase = SP == 2 ? kMargin + (tc ^ 1) : kMargin;
    const int ncols = (P.c_final - cbase + SP - 1) / SP, nrows = (P.H - kMargin - rbase + SP - 1) / SP;
    const int tiles_x = (ncols + TW - 1) / TW, tiles_y = (nrows + TH - 1) / TH;
    const int ntiles = (ncols > 0 && nrows > 0) ? tiles_x * tiles_y : 0;
    const int wg = (int)(blockIdx.x / (unsigned)ntypes), nwg = (int)(gridDim.x / (unsigned)ntypes);

    {   // the type's bank: 27 elements per thread, nine loads in flight at a time (a load -> store loop pays the memory latency 27 times)
        constexpr int NB = 217 * 128;
#pragma unroll 1
        for (int e0 = (int)threadIdx.x; e0 < NB; e0 += 9 * 1024) {
            float v[9];
#pragma unroll
            for (int u = 0; u < 9; u++) {
                const int e = e0 + 1024 * u, h = e >> 7, k = e & 127;
                v[u] = (e < NB && h < 216) ? P.bank[((size_t)h * ntypes + type) * kTapsPad + k] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 9; u++) {
                const int e = e0 + 1024 * u, h = e >> 7, k = e & 127, ch = k >> 4, l = k & 15;
                if (e < NB) sBank[h * 128 + (ch >> 2) * 64 + l * 4 + (ch & 3)] = v[u];
            }
        }
    }

    const int lane = threadIdx.x & 63, q = (int)(threadIdx.x >> 6);  // wave q <-> tile row q
    const int g = lane >> 4, l = lane & 15;
    // sample offset of tap k = 16 ch + l for the lane's pixel pair of step 0 (m0 = 2 g): window column SP m + tj lives in plane
    // tj % SP at index m + tj / SP; an odd index is read from the shifted copy at index - 1
    int off[8];
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
        const int k = 16 * ch + l;
        const int ti = k < kTaps ? k / 11 : 0, tj = k < kTaps ? k % 11 : 0;
        const int pl = tj % SP, idx = tj / SP, cp = idx & 1;
        const int reg = pl * 2 + cp;
        off[ch] = ti * RS + (reg == 0 ? REG0 : reg == 1 ? REG1 : reg == 2 ? REG2 : REG3) + (idx - cp) + 2 * g;
    }

    T regs[NLOAD];
    unsigned rh = 0xFFu, rh2 = 0xFFu;
    auto fetch = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int X0 = cbase + SP * TW * tx - 5, Y0 = rbase + SP * TH * ty - 5;
#pragma unroll
        for (int it = 0; it < NLOAD; it++) {
            const int e = min((int)threadIdx.x + 1024 * it, WW * WH - 1);
            const int wy = e / WW, wx = e - wy * WW;
            const int gy = min(max(Y0 + wy, 0), P.H - 1), gx = min(max(X0 + wx, 0), P.W - 1);
            regs[it] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gx];
        }
        const int r = rbase + SP * (TH * ty + q), c = cbase + SP * (TW * tx + lane);
        const bool in = r < P.H - kMargin && c < P.c_final;
        rh = in ? hash[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] : 0xFFu;
        rh2 = (in && c >= P.ov_begin && c < P.ov_end) ? P.hash2[(size_t)r * 16 + (c - P.ov_begin)] : 0xFFu;
    };
    auto stash = [&](int buf) {
        uint16_t* sT = buf ? sT1 : sT0;
#pragma unroll
        for (int it = 0; it < NLOAD; it++) {
            const int e = (int)threadIdx.x + 1024 * it;
            const int wy = e / WW, wx = e - wy * WW;
            if (e < WW * WH) {
                const uint16_t hv = __builtin_bit_cast(uint16_t, (_Float16)(float)regs[it]);     // exact: samples <= 1023
                const int pl = wx % SP, idx = wx / SP;
                uint16_t* row = sT + wy * RS;
                row[(pl ? REG2 : REG0) + idx] = hv;                   // copy 0
                if (idx > 0) row[(pl ? REG3 : REG1) + idx - 1] = hv;  // copy 1: shifted by one sample
            }
        }
        sHb[(buf * 2 + 0) * TH * TW + q * TW + lane] = (uint8_t)rh;
        sHb[(buf * 2 + 1) * TH * TW + q * TW + lane] = (uint8_t)rh2;
    };

    int tile = wg;
    if (tile < ntiles) { fetch(tile); stash(0); }
    lds_barrier();
    int cur = 0;
    const float negzero = -0.0f;
    for (; tile < ntiles; tile += nwg, cur ^= 1) {
        const int nxt = tile + nwg;
        if (nxt < ntiles) fetch(nxt);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const uint16_t* sT = cur ? sT1 : sT0;
        const uint8_t* sH = sHb + (cur * 2 + 0) * TH * TW + q * TW;
        const uint8_t* sH2 = sHb + (cur * 2 + 1) * TH * TW + q * TW;
        const int r = rbase + SP * (TH * ty + q);
        {
            const uint16_t* rowbase = sT + (SP * q) * RS;             // window row of image row r - 5
            const char* tap[8];
#pragma unroll
            for (int ch = 0; ch < 8; ch++) tap[ch] = reinterpret_cast<const char*>(rowbase + off[ch]);
#define RAISR_PAIR(p, ds) (*reinterpret_cast<const unsigned*>((p) + 16 * (ds)))      /* step ds: +8 samples */
            float keep = 0.0f;
            const bool anyB = sH2[lane] != 0xFFu;
            // the 16 buckets of the lane group's pixels, two per step (m = 8 ds + 2 g + e)
            unsigned hp[8];
#pragma unroll
            for (int ds = 0; ds < 8; ds++) hp[ds] = *reinterpret_cast<const uint16_t*>(sH + 8 * ds + 2 * g);
            // "virtual step" vs = 2 ds + e covers pixels m(vs, g); as in filter_phase the 16 virtual steps go in four groups
            // {j, j+4, j+8, j+12}: two DPP levels per accumulator, the four partial sums merged quad-wise, the last two levels,
            // the accept test and the keep-select once per group.  Steps ds = 0,2,4,6 feed groups 0 and 1, ds = 1,3,5,7 groups 2 and 3.
            const uint16_t* ctrrow = rowbase + 5 * RS + ((5 % SP) ? REG2 : REG0) + 5 / SP + 2 * g;
#pragma unroll
            for (int ph = 0; ph < 2; ph++) {
                float part0[4], part1[4];
#pragma unroll
                for (int mm = 0; mm < 4; mm++) {
                    const int ds = 2 * mm + ph;
                    const unsigned h0 = min(hp[ds] & 0xFFu, 216u), h1 = min(hp[ds] >> 8, 216u);
                    const float4 fa0 = *reinterpret_cast<const float4*>(sBank + h0 * 128u + (unsigned)l * 4u);
                    const float4 fb0 = *reinterpret_cast<const float4*>(sBank + h0 * 128u + 64u + (unsigned)l * 4u);
                    const float4 fa1 = *reinterpret_cast<const float4*>(sBank + h1 * 128u + (unsigned)l * 4u);
                    const float4 fb1 = *reinterpret_cast<const float4*>(sBank + h1 * 128u + 64u + (unsigned)l * 4u);
                    unsigned pw[8];
#pragma unroll
                    for (int ch = 0; ch < 8; ch++) pw[ch] = RAISR_PAIR(tap[ch], ds);
                    // acc = p[l] * f[l] is fma(p, f, -0) bit for bit; then the seven explicit fmadds of DotProdPatch
                    float a0 = fma_mix_lo(pw[0], fa0.x, negzero), a1 = fma_mix_hi(pw[0], fa1.x, negzero);
                    a0 = fma_mix_lo(pw[1], fa0.y, a0); a1 = fma_mix_hi(pw[1], fa1.y, a1);
                    a0 = fma_mix_lo(pw[2], fa0.z, a0); a1 = fma_mix_hi(pw[2], fa1.z, a1);
                    a0 = fma_mix_lo(pw[3], fa0.w, a0); a1 = fma_mix_hi(pw[3], fa1.w, a1);
                    a0 = fma_mix_lo(pw[4], fb0.x, a0); a1 = fma_mix_hi(pw[4], fb1.x, a1);
                    a0 = fma_mix_lo(pw[5], fb0.y, a0); a1 = fma_mix_hi(pw[5], fb1.y, a1);
                    a0 = fma_mix_lo(pw[6], fb0.z, a0); a1 = fma_mix_hi(pw[6], fb1.z, a1);
                    a0 = fma_mix_lo(pw[7], fb0.w, a0); a1 = fma_mix_hi(pw[7], fb1.w, a1);
                    a0 = a0 + row_ror<0x128>(a0); part0[mm] = a0 + row_ror<0x124>(a0);
                    a1 = a1 + row_ror<0x128>(a1); part1[mm] = a1 + row_ror<0x124>(a1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int j = 2 * ph + e;                          // group j: virtual steps j, j+4, j+8, j+12 <-> quads 0..3
                    float v = e ? part1[0] : part0[0];
                    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(e ? part1[1] : part0[1]), "s"(0x00f000f000f000f0ull));
                    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(e ? part1[2] : part0[2]), "s"(0x0f000f000f000f00ull));
                    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(e ? part1[3] : part0[3]), "s"(0xf000f000f000f000ull));
                    v = v + quad_perm<0x4e>(v);
                    v = v + quad_perm<0xb1>(v);
                    // this lane's quad (l >> 2) ends up with virtual step j + 4 (l >> 2): its centre pixel m = 8 (vs >> 1) + 2 g + (vs & 1)
                    const int vsq = j + 4 * (l >> 2);
                    float res = (float)__builtin_bit_cast(_Float16, ctrrow[8 * (vsq >> 1) + (vsq & 1)]);
                    if (v > P.lo && v < P.hi) res = v;
                    // lane (g, l) keeps virtual step l: in group j those are the lanes with (l & 3) == j
                    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(keep) : "v"(res), "s"(0x1111111111111111ull << j));
                }
            }
            if (__any(anyB)) {                                       // tail columns: AVX2 re-hash (keep-first-if-rejected)
#pragma unroll 1
                for (int ds = 0; ds < 8; ds++) {
                    const unsigned b0 = sH2[8 * ds + 2 * g], b1 = sH2[8 * ds + 2 * g + 1];
                    if (b0 == 0xFFu && b1 == 0xFFu) continue;
                    const unsigned h0 = min(b0, 216u), h1 = min(b1, 216u);
                    const float4 fa0 = *reinterpret_cast<const float4*>(sBank + h0 * 128u + (unsigned)l * 4u);
                    const float4 fb0 = *reinterpret_cast<const float4*>(sBank + h0 * 128u + 64u + (unsigned)l * 4u);
                    const float4 fa1 = *reinterpret_cast<const float4*>(sBank + h1 * 128u + (unsigned)l * 4u);
                    const float4 fb1 = *reinterpret_cast<const float4*>(sBank + h1 * 128u + 64u + (unsigned)l * 4u);
                    unsigned pw[8];
#pragma unroll
                    for (int ch = 0; ch < 8; ch++) pw[ch] = RAISR_PAIR(tap[ch], ds);
                    float a0 = fma_mix_lo(pw[0], fa0.x, negzero), a1 = fma_mix_hi(pw[0], fa1.x, negzero);
                    a0 = fma_mix_lo(pw[1], fa0.y, a0); a1 = fma_mix_hi(pw[1], fa1.y, a1);
                    a0 = fma_mix_lo(pw[2], fa0.z, a0); a1 = fma_mix_hi(pw[2], fa1.z, a1);
                    a0 = fma_mix_lo(pw[3], fa0.w, a0); a1 = fma_mix_hi(pw[3], fa1.w, a1);
                    a0 = fma_mix_lo(pw[4], fb0.x, a0); a1 = fma_mix_hi(pw[4], fb1.x, a1);
                    a0 = fma_mix_lo(pw[5], fb0.y, a0); a1 = fma_mix_hi(pw[5], fb1.y, a1);
                    a0 = fma_mix_lo(pw[6], fb0.z, a0); a1 = fma_mix_hi(pw[6], fb1.z, a1);
                    a0 = fma_mix_lo(pw[7], fb0.w, a0); a1 = fma_mix_hi(pw[7], fb1.w, a1);
                    const float v0 = tree16(a0), v1 = tree16(a1);
                    const uint16_t* cp = rowbase + 5 * RS + ((5 % SP) ? REG2 : REG0) + 5 / SP + 8 * ds + 2 * g;
                    if (l == 2 * ds && b0 != 0xFFu) {
                        if (v0 > P.lo && v0 < P.hi) keep = v0;
                        else if (P.randomness) keep = (float)__builtin_bit_cast(_Float16, cp[0]);
                    }
                    if (l == 2 * ds + 1 && b1 != 0xFFu) {
                        if (v1 > P.lo && v1 < P.hi) keep = v1;
                        else if (P.randomness) keep = (float)__builtin_bit_cast(_Float16, cp[1]);
                    }
                }
            }
#undef RAISR_PAIR
            const int m = 8 * (l >> 1) + 2 * g + (l & 1);
            const int c = cbase + SP * (TW * tx + m);
            if (r < P.H - kMargin && c < P.c_final) hr[(size_t)r * P.hr_pitch + c] = keep;
        }
        if (nxt < ntiles) stash(cur ^ 1);
        lds_barrier();
    }
}


// ------------------------------------------------------------------------------------------------
// k_filter_mfma: the filter stage on the matrix cores -- the NON-bit-exact "fast" mode of SURVEY 8(f) rank 4 / north_star
// ("MFMA only if the per-bucket filter application is reformulated as a dense batched GEMV").  Opt-in
// (raisr_hip_set_fast / RAISR_HIP_FAST=1), ratio 2, 8/10-bit, fp32 flavours; the buckets stay exact (certified hash stage of the
// split pipeline), only the 121-tap dot product (DotProdPatch_AVX512_32f, Raisr_AVX512.cpp:134-149) changes: coefficients
// rounded to binary16, products exact, sums in the MFMA's fp32 order instead of the 16-lane order.
//
// A bucket's dot product is a GEMV (pixels x taps) . (taps), which leaves the matrix cores idle; what fills them is
// redundancy: the nine buckets of one angle (strength x coherence) are the N dimension, padded to 16, and a pixel's
// row of D = A . B holds its candidate for each of them -- the wanted one is picked in the epilogue.  2 x 16 / 9
// x 192 / 121 = 5.6 x the useful flops, at 16 x the fp32-vector rate.  Per workgroup: a 128 x 32 pixel area (all four
// pixel types), its pixels counting-sorted in LDS by (type, angle) into 96 bins padded to 16-pixel groups; a wave
// walks a contiguous run of groups and keeps the B panel of the current bin (16 filters x 192 taps binary16, 24
// VGPRs) in registers.  K = 192: window row ti = 2 kb + (q >> 1), 16 slots per row (q & 1 picks the half), tap tj at
// slot tj + tc -- pixels of odd column type start their reads one sample early, which makes every A read even-aligned;
// four copies of the LR window (binary16, shifted by 0/2/4/6 samples) make it 16-byte aligned: one ds_read_b128 per
// lane and MFMA.  Slots without a tap have zero coefficients (their samples are finite: the whole row is staged).
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMfW = 128, kMfH = 32;                 // area (24 rows measure 6 % slower)
constexpr int kMfRS = 152, kMfRows = kMfH + 11;      // row stride (samples) and rows of one window copy
constexpr int kMfBins = 96;                          // (type, angle)
constexpr int kMfThreads = 1024, kMfWaves = kMfThreads / 64, kMfPer = kMfW * kMfH / kMfThreads;
constexpr int kMfSlots = kMfW * kMfH + kMfBins * 15 + 16;       // worst-case padded entries (5552, a multiple of 16)
constexpr int kMfCopy = kMfRows * kMfRS + 24;        // samples per window copy: 820 x 16 B, so copies sit 4 bank-quads apart
constexpr size_t kMfLds = (size_t)4 * kMfCopy * 2 + (size_t)kMfW * kMfH * 4 + (size_t)kMfSlots * 2 + (kMfBins + 104) * 4 + 352;
constexpr size_t kMfBankHalfs = (size_t)4 * 24 * 6 * 64 * 8;    // [type][angle][kb][lane][8]

__global__ __launch_bounds__(256) void k_build_mfma_bank(const float* __restrict__ bank, _Float16* __restrict__ out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= kMfBankHalfs) return;
    const unsigned e = i & 7u, lane = (i >> 3) & 63u, kb = (i >> 9) % 6u, bin = (i >> 9) / 6u;
    const unsigned angle = bin % 24u, type = bin / 24u;
    const unsigned n = lane & 15u, q = lane >> 4;
    const int ti = (int)(2u * kb + (q >> 1)), tj = (int)(8u * (q & 1u) + e) - (int)(type & 1u);
    float v = 0.0f;
    if (n < 9u && ti < 11 && tj >= 0 && tj < 11) v = bank[((size_t)(angle * 9u + n) * 4u + type) * kTapsPad + (unsigned)(ti * 11 + tj)];
    out[i] = (_Float16)v;
}

template <typename T, int PART = 0>
__global__ __launch_bounds__(kMfThreads) void k_filter_mfma(const T* __restrict__ lr, const uint8_t* __restrict__ hash, PassParams P,
                                                     const uint4* __restrict__ bankm, float* __restrict__ hr,
                                                     unsigned* __restrict__ fix_counters)
{
    extern __shared__ uint4 smem_mf[];
    uint16_t* sC = reinterpret_cast<uint16_t*>(smem_mf);             // [4][kMfCopy]: rows of kMfRS samples
    float* sRes = reinterpret_cast<float*>(sC + 4 * kMfCopy);        // [kMfH][128] filter stage output of the area (coalesced store at the end)
    uint16_t* sE = reinterpret_cast<uint16_t*>(sRes + kMfW * kMfH);  // sorted entries: pixel (12 bits) | column (4 bits)
    int* sCnt = reinterpret_cast<int*>(sE + kMfSlots);               // [96]
    int* sStart = sCnt + kMfBins;                                    // [97] first slot of a bin (multiples of 16)
    uint8_t* sGB = reinterpret_cast<uint8_t*>(sStart + 104);         // bin of a group

    if (fix_counters && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) fix_counters[0] = 0;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C0 = kMargin + kMfW * (int)blockIdx.x, R0 = kMargin + kMfH * (int)blockIdx.y;
    const int X0 = C0 - 6, Y0 = R0 - 5;                              // window origin: one sample early (see above)

    // the area's buckets first (their latency hides behind the staging)
    unsigned hreg[kMfPer];
#pragma unroll
    for (int u = 0; u < kMfPer; u++) {
        const int idx = tid + kMfThreads * u, py = idx >> 7, px = idx & 127;
        const int r = R0 + py, c = C0 + px;
        hreg[u] = (r < P.H - kMargin && c < P.c_final) ? hash[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] : 0xFFu;
    }
    for (int i = tid; i < kMfSlots / 2; i += kMfThreads) reinterpret_cast<unsigned*>(sE)[i] = 0xF000F000u;     // padding: pixel 0, column 15 (never stored)
    if (tid < kMfBins) sCnt[tid] = 0;
    {   // stage the window: sample pairs, four shifted copies
        constexpr int NP = kMfRows * (144 / 2);
#pragma unroll 1
        for (int e0 = tid; e0 < NP; e0 += 4 * kMfThreads) {
            T va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = min(e0 + kMfThreads * u, NP - 1), wy = e / 72, wx = 2 * (e - wy * 72);
                const int gy = min(max(Y0 + wy, 0), P.H - 1);
                const int gx0 = min(max(X0 + wx, 0), P.W - 1), gx1 = min(max(X0 + wx + 1, 0), P.W - 1);
                va[u] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gx0];
                vb[u] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gx1];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + kMfThreads * u;
                if (e < NP) {
                    const int wy = e / 72, wx = 2 * (e - wy * 72);
                    const unsigned pair = (unsigned)__builtin_bit_cast(uint16_t, (_Float16)(float)va[u]) |
                                          ((unsigned)__builtin_bit_cast(uint16_t, (_Float16)(float)vb[u]) << 16);
                    uint16_t* d = sC + wy * kMfRS + 8 + wx;
#pragma unroll
                    for (int s = 0; s < 4; s++) *reinterpret_cast<unsigned*>(d + s * kMfCopy - 2 * s) = pair;
                }
            }
        }
    }
    lds_barrier();
    // counting sort by (type, angle): rank inside the bin from an LDS atomic
    int rank[kMfPer];
#pragma unroll
    for (int u = 0; u < kMfPer; u++) {
        const int idx = tid + kMfThreads * u, py = idx >> 7, px = idx & 127;
        const unsigned h = hreg[u];
        const int type = ((py + 1) & 1) * 2 + ((px + 1) & 1);
        rank[u] = h < 216u ? atomicAdd(&sCnt[type * 24 + (int)(h / 9u)], 1) : -1;
    }
    lds_barrier();
    if (tid < 64) {                                                   // exclusive scan of the padded counts
        const int c0 = (sCnt[tid] + 15) & ~15, c1 = tid < kMfBins - 64 ? (sCnt[tid + 64] + 15) & ~15 : 0;
        int inc0 = c0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc0, d); if (lane >= d) inc0 += t; }
        const int tot0 = __shfl(inc0, 63);
        int inc1 = c1;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up(inc1, d); if (lane >= d) inc1 += t; }
        sStart[tid] = inc0 - c0;
        if (tid < kMfBins - 64) sStart[tid + 64] = tot0 + inc1 - c1;
        if (tid == kMfBins - 64 - 1) sStart[kMfBins] = tot0 + inc1;
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < kMfPer; u++) {
        const int idx = tid + kMfThreads * u, py = idx >> 7, px = idx & 127;
        const unsigned h = hreg[u];
        if (rank[u] >= 0) {
            const int type = ((py + 1) & 1) * 2 + ((px + 1) & 1);
            sE[sStart[type * 24 + (int)(h / 9u)] + rank[u]] = (uint16_t)((unsigned)idx | ((h % 9u) << 12));
        }
    }
    if (tid < kMfBins) {
        const int g0 = sStart[tid] >> 4, g1 = sStart[tid + 1] >> 4;
        for (int g = g0; g < g1; g++) sGB[g] = (uint8_t)tid;
    }
    lds_barrier();

    if (PART == 1) return;                                           // profiling aid: staging + sort only
    const int G = sStart[kMfBins] >> 4;
    const int gbeg = (G * wave) / kMfWaves, gend = (G * (wave + 1)) / kMfWaves;
    const int q = lane >> 4, col = lane & 15;
    int curbin = -1;
    f16x8 B[6];
#pragma unroll
    for (int kb = 0; kb < 6; kb++) B[kb] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int g = gbeg; g < gend; g++) {
        const int bin = __builtin_amdgcn_readfirstlane((int)sGB[g]);
        if (bin != curbin) {
            curbin = bin;
#pragma unroll
            for (int kb = 0; kb < 6; kb++) B[kb] = __builtin_bit_cast(f16x8, bankm[(unsigned)(bin * 6 + kb) * 64u + (unsigned)lane]);
        }
        const unsigned e = sE[g * 16 + col];
        const int idx = (int)(e & 4095u), py = idx >> 7, px = idx & 127;
        const int x0e = (px + 1) & ~1, s = (x0e >> 1) & 3;
        const uint16_t* a0 = sC + s * kMfCopy + (py + (q >> 1)) * kMfRS + 8 + x0e - 2 * s + 8 * (q & 1);
        f16x8 A[6];
        if (PART == 2) a0 = sC + lane * 8;                           // profiling aid: conflict-free A reads (wrong data)
#pragma unroll
        for (int kb = 0; kb < 6; kb++) A[kb] = *reinterpret_cast<const f16x8*>(a0 + kb * 2 * kMfRS);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < 6; kb++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kb], B[kb], acc, 0, 0, 0);
        // row 4 q + j of D belongs to entry 4 q + j of the group; the lane whose column is that pixel's bucket stores it
        const uint2 ee = *reinterpret_cast<const uint2*>(sE + g * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned ej = ((j < 2 ? ee.x : ee.y) >> (16 * (j & 1))) & 0xFFFFu;
            if ((int)(ej >> 12) == col && col < 9) sRes[ej & 4095u] = acc[j];
        }
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < kMfPer; u++) {
        const int idx = tid + kMfThreads * u, py = idx >> 7, px = idx & 127;
        if (hreg[u] < 216u) {
            const float v = sRes[idx];
            float res = (float)__builtin_bit_cast(_Float16, sC[(py + 5) * kMfRS + 8 + px + 6]);
            if (v > P.lo && v < P.hi) res = v;                       // accept test, Raisr.cpp:1196-1200
            hr[(size_t)(R0 + py) * P.hr_pitch + (C0 + px)] = res;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k_blend (CountOfBitsChanged): CTCountOfBitsChangedSegment_AVX256_32f, Raisr_AVX256.cpp:68-166,
// plus the border policy of processSegment (Raisr.cpp:999-1028,1252-1265): row 0, row H-1, col 0,
// col W-1 keep the unclamped LR value.
// ------------------------------------------------------------------------------------------------
template <typename TOut>
__global__ __launch_bounds__(256) void k_blend(const TOut* __restrict__ lr, const float* __restrict__ hr,
                                               PassParams P, TOut* __restrict__ out, int out_pitch)
{
    // tile 64 x 16 output pixels; wave w owns rows [4w, 4w+4), lane = column.  LR/HR tiles with a
    // 1-px halo are staged in LDS (HR := LR outside the filtered zone, Raisr.cpp:1035).
    constexpr int TW = 64, TH = 16, LW = TW + 2, LH = TH + 2;
    __shared__ float sL[LH * LW];
    __shared__ float sH[LH * LW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = bx * TW, r0 = by * TH;
    {   // wave w sweeps columns [0,64) of tile rows w, w+4, ...; the two right-hand halo columns go to the first 36 threads.
        // All LR and HR loads of a thread are in flight before the first LDS write.
        constexpr int NM = (LH + 3) / 4, REM = LW - 64;
        static_assert(LH * REM <= 256, "halo columns fit one sweep");
        const int wu = __builtin_amdgcn_readfirstlane(w);
        __builtin_assume(wu >= 0 && wu < 4);
        const int gxm = min(max(c0 - 1 + lane, 0), P.W - 1);
        const bool inxm = gxm >= kMargin && gxm < P.c_final;
        TOut lv[NM + 1];
        float hv[NM + 1];
        bool inz[NM + 1];
#pragma unroll
        for (int it = 0; it < NM; it++) {
            const int gy = min(max(r0 - 1 + min(wu + 4 * it, LH - 1), 0), P.H - 1);
            lv[it] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gxm];
            inz[it] = inxm && gy >= kMargin && gy < P.H - kMargin;
            hv[it] = inz[it] ? hr[(unsigned)gy * (unsigned)P.hr_pitch + (unsigned)gxm] : 0.0f;
        }
        const int rty = min((int)(threadIdx.x / REM), LH - 1), rtx = 64 + (int)(threadIdx.x % REM);
        {
            const int gy = min(max(r0 - 1 + rty, 0), P.H - 1), gx = min(max(c0 - 1 + rtx, 0), P.W - 1);
            lv[NM] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gx];
            inz[NM] = gy >= kMargin && gy < P.H - kMargin && gx >= kMargin && gx < P.c_final;
            hv[NM] = inz[NM] ? hr[(unsigned)gy * (unsigned)P.hr_pitch + (unsigned)gx] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NM; it++) {
            const int ty = wu + 4 * it;
            const float L = (float)lv[it];
            if (ty < LH) {
                sL[ty * LW + lane] = L;
                sH[ty * LW + lane] = inz[it] ? hv[it] : L;                          // HR := LR outside the filtered zone
            }
        }
        if (threadIdx.x < LH * REM) {
            const float L = (float)lv[NM];
            sL[rty * LW + rtx] = L;
            sH[rty * LW + rtx] = inz[NM] ? hv[NM] : L;
        }
    }
    __syncthreads();
    const int x = c0 + lane;
    if (x >= P.W) return;
    // sliding 3-row window down the wave's 4 rows
    float l[3][3], h[3][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            l[i + 1][j] = sL[(4 * w + i) * LW + lane + j];
            h[i + 1][j] = sH[(4 * w + i) * LW + lane + j];
        }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int y = r0 + 4 * w + rr;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            l[0][j] = l[1][j]; l[1][j] = l[2][j]; h[0][j] = h[1][j]; h[1][j] = h[2][j];
            l[2][j] = sL[(4 * w + rr + 2) * LW + lane + j];
            h[2][j] = sH[(4 * w + rr + 2) * LW + lane + j];
        }
        if (y >= P.H) break;
        const float Lc = l[1][1], Hc = h[1][1];
        int iv;
        if (x == 0 || y == 0 || x == P.W - 1 || y == P.H - 1) {
            iv = (int)Lc;                                       // unclamped LR copy
        } else {
            int hd = 0;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (i == 1 && j == 1) continue;
                    hd += ((l[i][j] < Lc) != (h[i][j] < Hc));
                }
            const float weight = (float)hd * 0.125f;           // hd / 8.0f exactly
            const float w2 = 1.0f - weight;
            float val = (weight * Lc) + (w2 * Hc);
            val = val + 0.5f;
            const float fl = __builtin_floorf(val);
            iv = (fl >= -2147483648.0f && fl < 2147483648.0f) ? (int)fl : (int)0x80000000;
            iv = max(min(iv, P.ihi), P.ilo);
        }
        out[(size_t)y * out_pitch + x] = (TOut)iv;
    }
}

// ------------------------------------------------------------------------------------------------
// k_blend_rand (BlendingMode Randomness): CTRandomness_AVX512_32f (Raisr_AVX512.cpp:19-35) + the inline
// blend of processSegment (Raisr.cpp:1203-1242).  Only the filtered pixels are blended; every other
// pixel is the unclamped LR copy, and the W-6-c_final pixels [c_final, W-6) of row H-7 are never
// written by the reference -- they keep whatever the output buffer held.
// The fp16 pipeline promotes to fp32 for this blend (Raisr.cpp:1224-1230), so one kernel serves both.
// ------------------------------------------------------------------------------------------------
template <typename TOut, bool HR16>
__global__ __launch_bounds__(256) void k_blend_rand(const TOut* __restrict__ lr, const void* __restrict__ hr,
                                                    PassParams P, TOut* __restrict__ out, int out_pitch)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.H) return;
    const TOut lc = lr[(size_t)y * P.lr_pitch + x];
    const bool zone = y >= kMargin && y < P.H - kMargin && x >= kMargin && x < P.c_final;
    if (!zone) {
        const bool untouched = P.H >= 2 * kMargin + 1 && y == P.H - kMargin - 1 && x >= P.c_final && x < P.W - kMargin;
        if (!untouched) out[(size_t)y * out_pitch + x] = (TOut)lc;
        return;
    }
    const float Lc = (float)lc;
    float cur;
    if (HR16) cur = (float)__builtin_bit_cast(_Float16, ((const uint16_t*)hr)[(size_t)y * P.hr_pitch + x]);
    else cur = ((const float*)hr)[(size_t)y * P.hr_pitch + x];
    int census = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            if (dx == 0 && dy == 0) continue;
            census += ((float)lr[(size_t)(y + dy) * P.lr_pitch + x + dx] < Lc);
        }
    const float weight = (float)census * 0.125f;                // census / 8.0f exactly
    float val = (weight * cur) + ((1.0f - weight) * Lc);
    val = val + 0.5f;
    const float cl = val < P.lo ? P.lo : (val > P.hi ? P.hi : val);
    out[(size_t)y * out_pitch + x] = (TOut)(int)cl;
}

// ------------------------------------------------------------------------------------------------
// Host side of the C ABI
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

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
    bool fused = true;                         // one k_hashfilter launch per pass instead of k_hash + k_filter (RAISR_HIP_FUSED=0)
    bool certify = true;                       // certified hash stage (k_hashfilter_ac / k_hash_ac); RAISR_HIP_CERTIFY=0 keeps the all-exact kernels
    bool split = false;                        // certified hash stage and filter stage as separate launches (RAISR_HIP_SPLIT=1)
    int fast = 0;                              // NON-bit-exact fast mode (raisr_hip_set_fast / RAISR_HIP_FAST): 1 = exact buckets, filter stage on the matrix cores; 2 = also keeps the approximate tensor's bucket where it is not certified
    bool lds_filter = true;                    // split pipeline: filter stage with the bank in LDS (RAISR_HIP_LDS_FILTER=0: k_filter)
    int n_cus = 256;                           // persistent k_filter_lds16 grid: one workgroup per CU (multiple of 4)
    float* d_gauss = nullptr;                  // GaussW::wT on the device (16-lane exact tensor of the worklist)
    FixLists fix{};                            // split pipeline: worklists of the certified hash stage (sized at configure)
    size_t fix_tiles = 0;
    int cert_check = 0;                        // tests: every pixel also takes the exact path, certified buckets are compared
    unsigned* d_cert_stats = nullptr;          // {uncertain, certified-but-wrong, zone pixels}, accumulated while non-null
    SepW sep{};
    int keep_hash_plane = 0;                   // fused kernel also writes the hash plane (set by raisr_hip_debug_read_stage users)
    raisr_hip_config cfg{};
    int blending = RAISR_HIP_BLEND_COUNT;      // per-call BlendingMode (RNLProcess argument)
    bool configured = false;
    ModelDev model[2];
    // shared small tables
    uint2* d_tab14 = nullptr;
    uint16_t* d_lut = nullptr;
    uint16_t* d_tab16 = nullptr;                // rcpph T, rsqrtph T0, T1
    GaussW16 gauss16{};
    // scratch planes
    void* d_lr[2] = {nullptr, nullptr};         // LR plane per pass, sample type (u8 for 8-bit content, else u16)
    uint8_t* d_hash[2] = {nullptr, nullptr};    // first hash per pixel
    uint8_t* d_hash2[2] = {nullptr, nullptr};   // [H][16] second hash of the tail (overlap) columns
    float* d_hr[2] = {nullptr, nullptr};
    void* d_mid = nullptr;                      // two-pass intermediate (sample type), only when the passes differ in size
    int passW[2] = {0, 0}, passH[2] = {0, 0};
    GaussW gauss{};
    // device staging for raisr_hip_process_host
    void* d_stage = nullptr; size_t d_stage_bytes = 0;
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
    const int gx = gcd_int(sw, dw), gy = gcd_int(sh, dh);
    R.Sx = sw / gx; R.Dx = dw / gx; R.Sy = sh / gy; R.Dy = dh / gy;
    R.tie_even = tie == RAISR_HIP_TIE_HALF_EVEN;
    return R;
}

template <typename TIn, typename TOut>
void launch_resize(raisr_hip_ctx* c, hipStream_t s, const void* src, void* dst, const ResizeParams& R, const char* name)
{
    int slot;
    timer_begin(c, name, s, slot);
    if (R.dw == 2 * R.sw && R.dh == 2 * R.sh) {
        dim3 grid(((R.dw + 3) / 4 + 63) / 64, (R.dh + 3) / 4);
        hipLaunchKernelGGL((k_resize2x<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else if (R.dw == R.sw && R.dh == R.sh) {
        dim3 grid((R.dw + 63) / 64, (R.dh + 3) / 4);
        hipLaunchKernelGGL((k_copy<TIn, TOut>), grid, dim3(256), 0, s, (const TIn*)src, (TOut*)dst, R);
    } else {
        dim3 grid((R.dw + 63) / 64, (R.dh + 3) / 4);
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
    P.tab14 = c->d_tab14;
    P.lut_legacy = c->d_lut;
    P.zero_bucket[0] = m.zero_bucket[0]; P.zero_bucket[1] = m.zero_bucket[1];
    P.gauss_dev = c->d_gauss;
    return P;
}

// one RAISR pass on an LR plane already resident in c->d_lr[pass]
template <typename TOut>
void run_pass(raisr_hip_ctx* c, hipStream_t s, int pass, void* out, int out_pitch_elems)
{
    const int W = c->passW[pass], H = c->passH[pass];
    PassParams P = make_pass(c, pass, W, H);
    int slot;
    if (P.c_final > kMargin && H > 2 * kMargin) {
        constexpr int R = 4;        // rows per lane: 3..6 measure the same within noise, 8 is slower (occupancy)
        dim3 gh((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 4 * R - 1) / (4 * R));
        dim3 gf((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 15) / 16);
        const bool avx2all = !(P.a_end > P.a_begin);        // asm=avx2: no 16-wide chunks at all
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
            hipLaunchKernelGGL((k_hash_ac<TOut>), dim3(npers), dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->sep, F, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            if (c->fast < 2) {
            timer_begin(c, "k_fix", s, slot);
            hipLaunchKernelGGL((k_fix_sparse<TOut>), dim3((ntiles + 3u) / 4u), dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, F, c->d_hash[pass], c->d_hash2[pass]);
            const unsigned nd = (unsigned)(gf.x * gf.y < 2048u ? gf.x * gf.y : 2048u);
            if (!avx2all)
                hipLaunchKernelGGL((k_fix_dense<TOut, false>), dim3(nd), dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, F, c->d_hash[pass], c->d_hash2[pass]);
            else
                hipLaunchKernelGGL((k_fix_dense<TOut, true>), dim3(nd), dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, F, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            }
            if (c->fast && c->model[pass].bank_mfma) {             // the panels are allocated by configure / set_fast (errors reported there)
                ModelDev& m = c->model[pass];
                if (!m.bank_mfma_valid) {
                    hipLaunchKernelGGL(k_build_mfma_bank, dim3((unsigned)((kMfBankHalfs + 255) / 256)), dim3(256), 0, s, P.bank, m.bank_mfma);
                    m.bank_mfma_valid = true;
                }
                {
                    dim3 gm((P.c_final - kMargin + kMfW - 1) / kMfW, (H - 2 * kMargin + kMfH - 1) / kMfH);
                    timer_begin(c, "k_filter_mfma", s, slot);
                    static const int mpart = getenv("RAISR_HIP_MF_PART") ? atoi(getenv("RAISR_HIP_MF_PART")) : 0;
                    if (mpart == 1) hipLaunchKernelGGL((k_filter_mfma<TOut, 1>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    else if (mpart == 2) hipLaunchKernelGGL((k_filter_mfma<TOut, 2>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    else
                    hipLaunchKernelGGL((k_filter_mfma<TOut>), gm, dim3(kMfThreads), kMfLds, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, (const uint4*)m.bank_mfma, c->d_hr[pass], F.counters);
                    timer_end(c, s, slot);
                }
            } else if (c->lds_filter && c->cfg.bits <= 10) {      // samples above 10 bits are not exact in binary16: k_filter
                const bool sp2 = P.pixel_types == 4;
                const int wh = sp2 ? 41 : 26;
                const size_t sh16 = (size_t)217 * 128 * 4 + 2 * (size_t)wh * (sp2 ? 286 : 154) * 2 + 4 * 1024;
                timer_begin(c, "k_filter_lds16", s, slot);
                if (sp2) hipLaunchKernelGGL((k_filter_lds16<TOut, 2>), dim3(c->n_cus), dim3(1024), sh16, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                else hipLaunchKernelGGL((k_filter_lds16<TOut, 1>), dim3(c->n_cus), dim3(1024), sh16, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                timer_end(c, s, slot);
            } else {
                timer_begin(c, "k_filter", s, slot);
                hipLaunchKernelGGL((k_filter<TOut>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass], F.counters);
                timer_end(c, s, slot);
            }
        } else if (c->fused && c->certify) {
            P.write_hash = c->keep_hash_plane;
            P.cert_stats = c->d_cert_stats;
            P.cert_check = c->cert_check;
            timer_begin(c, "k_hashfilter_ac", s, slot);
            static const int part = getenv("RAISR_HIP_AC_PART") ? atoi(getenv("RAISR_HIP_AC_PART")) : 0;
            if (part == 1)
                hipLaunchKernelGGL((k_hashfilter_ac<TOut, 1>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
            else if (part == 2)
                hipLaunchKernelGGL((k_hashfilter_ac<TOut, 2>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
            else
                hipLaunchKernelGGL((k_hashfilter_ac<TOut, 0>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->sep, c->d_hash[pass], c->d_hr[pass]);
            timer_end(c, s, slot);
        } else if (c->fused) {
            P.write_hash = c->keep_hash_plane;
            timer_begin(c, "k_hashfilter", s, slot);
            if (!avx2all)
                hipLaunchKernelGGL((k_hashfilter<TOut, false>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->d_hash[pass], c->d_hr[pass]);
            else
                hipLaunchKernelGGL((k_hashfilter<TOut, true>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->d_hash[pass], c->d_hr[pass]);
            timer_end(c, s, slot);
        } else {
            timer_begin(c, "k_hash", s, slot);
            if (!avx2all)
                hipLaunchKernelGGL((k_hash<R, TOut, false>), gh, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->d_hash[pass], c->d_hash2[pass]);
            else
                hipLaunchKernelGGL((k_hash<R, TOut, true>), gh, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, c->gauss, c->d_hash[pass], c->d_hash2[pass]);
            timer_end(c, s, slot);
            timer_begin(c, "k_filter", s, slot);
            hipLaunchKernelGGL((k_filter<TOut>), gf, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, c->d_hr[pass]);
            timer_end(c, s, slot);
        }
    }
    if (P.randomness) {
        dim3 gb((W + 63) / 64, (H + 3) / 4);
        timer_begin(c, "k_blend_rand", s, slot);
        hipLaunchKernelGGL((k_blend_rand<TOut, false>), gb, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const void*)c->d_hr[pass], P, (TOut*)out, out_pitch_elems);
        timer_end(c, s, slot);
        return;
    }
    dim3 gb((W + 63) / 64, (H + 15) / 16);
    timer_begin(c, "k_blend", s, slot);
    hipLaunchKernelGGL((k_blend<TOut>), gb, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const float*)c->d_hr[pass], P, (TOut*)out, out_pitch_elems);
    timer_end(c, s, slot);
}

// one pass of the AVX512-FP16-exact pipeline (binary16 arithmetic)
template <typename TOut>
void run_pass16(raisr_hip_ctx* c, hipStream_t s, int pass, void* out, int out_pitch_elems)
{
    const int W = c->passW[pass], H = c->passH[pass];
    PassParams P = make_pass(c, pass, W, H);
    const ModelDev& m = c->model[pass];
    Pass16 Q{};
    const int rows = m.h.hashkeys * m.h.pixel_types;
    Q.bank16 = (const uint32_t*)((const char*)m.blob + kBlobHeader + blob_f32_bytes(rows));
    Q.bank16_bytes = (int)blob_f16_bytes(rows);
    Q.tab16 = c->d_tab16;
    Q.qangle = m.h.qangle16; Q.qs0 = m.h.qstr16[0]; Q.qs1 = m.h.qstr16[1]; Q.qc0 = m.h.qcoh16[0]; Q.qc1 = m.h.qcoh16[1];
    {   // NF_8 / NF_10 (Raisr_globals.h:208-209; Raisr_AVX512FP16.cpp:146-151)
        const float maxv = c->cfg.bits == 8 ? 255.0f : 1023.0f;
        volatile float nf = 1.0f / (maxv * maxv * 2.0f * 2.0f);
        Q.nf = nf;
    }
    Q.c_avx = (W - 1) - ((W - 1) % 32) + 1;
    int slot;
    if (P.c_final > kMargin && H > 2 * kMargin) {
        dim3 gh((P.c_final - kMargin + 63) / 64, (H - 2 * kMargin + 15) / 16);
        if (c->fused) {
            P.write_hash = c->keep_hash_plane;
            timer_begin(c, "k_hashfilter16", s, slot);
            hipLaunchKernelGGL((k_hashfilter16<TOut>), gh, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, Q, c->gauss16, c->d_hash[pass], (uint16_t*)c->d_hr[pass]);
            timer_end(c, s, slot);
        } else {
            timer_begin(c, "k_hash16", s, slot);
            hipLaunchKernelGGL((k_hash16<4, TOut>), gh, dim3(256), 0, s, (const TOut*)c->d_lr[pass], P, Q, c->gauss16, c->d_hash[pass]);
            timer_end(c, s, slot);
            timer_begin(c, "k_filter16", s, slot);
            hipLaunchKernelGGL((k_filter16<TOut>), gh, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const uint8_t*)c->d_hash[pass], P, Q, (uint16_t*)c->d_hr[pass]);
            timer_end(c, s, slot);
        }
    }
    if (P.randomness) {
        dim3 gr((W + 63) / 64, (H + 3) / 4);
        timer_begin(c, "k_blend_rand", s, slot);
        hipLaunchKernelGGL((k_blend_rand<TOut, true>), gr, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const void*)c->d_hr[pass], P, (TOut*)out, out_pitch_elems);
        timer_end(c, s, slot);
        return;
    }
    dim3 gb((W + 63) / 64, (H + 15) / 16);
    timer_begin(c, "k_blend16", s, slot);
    hipLaunchKernelGGL((k_blend16<TOut>), gb, dim3(256), 0, s, (const TOut*)c->d_lr[pass], (const uint16_t*)c->d_hr[pass], P, Q, (TOut*)out, out_pitch_elems);
    timer_end(c, s, slot);
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
    if (c->fix.counters) (void)hipFree(c->fix.counters);
    if (c->fix.counts) (void)hipFree(c->fix.counts);
    if (c->fix.sparse) (void)hipFree(c->fix.sparse);
    if (c->fix.dense) (void)hipFree(c->fix.dense);
    if (c->fix.cert_mask) (void)hipFree(c->fix.cert_mask);
    c->fix = FixLists{};
    c->fix_tiles = 0;
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
    if (const char* e = getenv("RAISR_HIP_FAST")) { const int v = atoi(e); c->fast = v < 0 ? 0 : (v > 2 ? 2 : v); }         // NON-bit-exact fast mode (see raisr_hip_set_fast)
    if (const char* e = getenv("RAISR_HIP_SPLIT")) c->split = atoi(e) != 0;       // A/B switch: 1 = k_hash_ac + filter kernel
    if (const char* e = getenv("RAISR_HIP_LDS_FILTER")) c->lds_filter = atoi(e) != 0;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount >= 4) c->n_cus = prop.multiProcessorCount & ~3;
        // k_filter_lds16 declares ~160 KB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_mfma<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint8_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint8_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_filter_lds16<uint16_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
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
    std::vector<uint16_t> t16(3072);
    for (int i = 0; i < 1024; i++) { t16[i] = X86_RCPPH_T[i]; t16[1024 + i] = X86_RSQRTPH_T0[i]; t16[2048 + i] = X86_RSQRTPH_T1[i]; }
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
    if (c->d_tab14) (void)hipFree(c->d_tab14);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_tab16) (void)hipFree(c->d_tab16);
    if (c->d_cert_stats) (void)hipFree(c->d_cert_stats);
    if (c->ev_chroma) (void)hipEventDestroy(c->ev_chroma);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->ev_comp) (void)hipEventDestroy(c->ev_comp);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
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
    return compute_zero_buckets(c, pass_index);
}

static bool fast_mode_supported(const raisr_hip_config* cfg)
{
    return cfg->use_pixel_type && cfg->bits <= 10 && cfg->hash_variant != RAISR_HIP_HASH_FP16;
}

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

int raisr_hip_set_fast(raisr_hip_ctx* c, int on)
{
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
    if (c->fast && !fast_mode_supported(cfg))
        return fail(RAISR_HIP_EINVAL, "fast mode (matrix-core filter stage) supports ratio 2, 8/10-bit content and the fp32 flavours only");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_scratch(c);
    c->cfg = *cfg;
    c->gauss = make_gauss(cfg->bits);
    c->sep = make_sep(c->gauss);
    HIP_TRY(hipMemcpy(c->d_gauss, &c->gauss, sizeof(GaussW), hipMemcpyHostToDevice));
    const bool mode2 = cfg->passes == 2 && cfg->two_pass_mode == 2;
    c->passW[0] = mode2 ? cfg->in_width : cfg->out_width;
    c->passH[0] = mode2 ? cfg->in_height : cfg->out_height;
    c->passW[1] = cfg->out_width; c->passH[1] = cfg->out_height;
    const size_t bps = cfg->bits == 8 ? 1 : 2;
    for (int p = 0; p < cfg->passes; p++) {
        const size_t n = (size_t)c->passW[p] * c->passH[p];
        if (hipMalloc((void**)&c->d_lr[p], n * bps) != hipSuccess ||
            hipMalloc((void**)&c->d_hash[p], n) != hipSuccess ||
            hipMalloc((void**)&c->d_hash2[p], (size_t)c->passH[p] * 16) != hipSuccess ||
            hipMalloc((void**)&c->d_hr[p], n * sizeof(float)) != hipSuccess) {
            free_scratch(c);
            return fail(RAISR_HIP_ENOMEM, "scratch plane alloc");
        }
    }
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
    }
    if (cfg->passes == 2) {
        // pixels the Randomness pass never writes stay 0 in the intermediate (the reference leaves heap garbage there)
        HIP_TRY(hipMemsetAsync(c->d_lr[1], 0, (size_t)c->passW[1] * c->passH[1] * bps, c->stream));
        if (c->passW[0] != c->passW[1] || c->passH[0] != c->passH[1]) {
            const size_t n = (size_t)c->passW[0] * c->passH[0];
            if (hipMalloc((void**)&c->d_mid, n * bps) != hipSuccess) { free_scratch(c); return fail(RAISR_HIP_ENOMEM, "intermediate alloc"); }
            HIP_TRY(hipMemsetAsync(c->d_mid, 0, n * bps, c->stream));
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

int raisr_hip_set_blending(raisr_hip_ctx* c, int blending)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    if (blending != RAISR_HIP_BLEND_COUNT && blending != RAISR_HIP_BLEND_RANDOMNESS)
        return fail(RAISR_HIP_EINVAL, "blending must be 1 (Randomness) or 2 (CountOfBitsChanged)");
    c->blending = blending;
    return RAISR_HIP_OK;
}

int raisr_hip_process_y_device(raisr_hip_ctx* c, const void* d_in, size_t in_pitch, void* d_out, size_t out_pitch, void* stream)
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
    auto job = [&](auto tag) {
        using T = decltype(tag);
        ResizeParams R0 = make_resize(g.in_width, g.in_height, ipe, c->passW[0], c->passH[0], c->passW[0], g.tie_rule);
        launch_resize<T, T>(c, s, d_in, c->d_lr[0], R0, "k_resize");
        if (g.passes == 1) {
            if (fp16) run_pass16<T>(c, s, 0, d_out, ope); else run_pass<T>(c, s, 0, d_out, ope);
            return;
        }
        // pass 1 writes the integer intermediate (Raisr.cpp:927-934).  When both passes run at output size
        // (mode 1) the intermediate IS pass 2's LR plane; in mode 2 it is upscaled now (Raisr.cpp:945-975).
        void* mid = same ? c->d_lr[1] : c->d_mid;
        if (fp16) run_pass16<T>(c, s, 0, mid, c->passW[0]); else run_pass<T>(c, s, 0, mid, c->passW[0]);
        if (!same) {
            ResizeParams R1 = make_resize(c->passW[0], c->passH[0], c->passW[0], c->passW[1], c->passH[1], c->passW[1], g.tie_rule);
            launch_resize<T, T>(c, s, c->d_mid, c->d_lr[1], R1, "k_resize");
        }
        if (fp16) run_pass16<T>(c, s, 1, d_out, ope); else run_pass<T>(c, s, 1, d_out, ope);
    };
    if (bps == 1) job(uint8_t{}); else job(uint16_t{});
    HIP_TRY(hipGetLastError());
    return RAISR_HIP_OK;
}

int raisr_hip_resize_plane_device(raisr_hip_ctx* c, const void* d_src, int sw, int sh, size_t spitch,
                                  void* d_dst, int dw, int dh, size_t dpitch, int bits, void* stream)
{
    if (!c || !d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return fail(RAISR_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const int bps = bits == 8 ? 1 : 2;
    ResizeParams R = make_resize(sw, sh, (int)(spitch / bps), dw, dh, (int)(dpitch / bps), c->configured ? c->cfg.tie_rule : 0);
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
    if (!c || !d_in_u || !d_in_v || !d_out_u || !d_out_v) return fail(RAISR_HIP_EINVAL, "null plane");
    if (!c->configured) return fail(RAISR_HIP_ESTATE, "configure first");
    int rc = raisr_hip_process_y_device(c, d_in_y, in_y_pitch, d_out_y, out_y_pitch, stream);
    if (rc) return rc;
    rc = raisr_hip_resize_plane_device(c, d_in_u, cin_w, cin_h, in_c_pitch, d_out_u, cout_w, cout_h, out_c_pitch, c->cfg.bits, stream);
    if (rc) return rc;
    return raisr_hip_resize_plane_device(c, d_in_v, cin_w, cin_h, in_c_pitch, d_out_v, cout_w, cout_h, out_c_pitch, c->cfg.bits, stream);
}

int raisr_hip_synchronize(raisr_hip_ctx* c)
{
    if (!c) return fail(RAISR_HIP_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (c->up) {                               // shared streams: wait for this context's last frame only
        if (c->done_pending) { HIP_TRY(hipEventSynchronize(c->ev_done)); c->done_pending = false; }
        return RAISR_HIP_OK;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream2));
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
    }
    char* d = (char*)c->d_stage;
    if (c->up && !rows) {
        // stage-ordered pipeline of the stream ring: uploads on the shared upload stream, kernels (Y, then the two cheap chroma
        // upscales) on the compute stream, one download on the shared download stream; events carry the dependencies on the device
        const size_t irow = (size_t)g.in_width * bps, orow = (size_t)g.out_width * bps;
        const size_t cirow = (size_t)cin_w * bps, crow = (size_t)cout_w * bps;
        hipStream_t s = c->stream;
        HIP_TRY(copy_plane(d, irow, in_y, in_y_pitch, irow, g.in_height, hipMemcpyHostToDevice, c->up));
        if (c->blending == RAISR_HIP_BLEND_RANDOMNESS)
            HIP_TRY(copy_plane(d + off_oy, orow, out_y, out_y_pitch, orow, g.out_height, hipMemcpyHostToDevice, c->up));
        if (chroma) {
            HIP_TRY(copy_plane(d + off_iu, cirow, in_u, in_u_pitch, cirow, cin_h, hipMemcpyHostToDevice, c->up));
            HIP_TRY(copy_plane(d + off_iv, cirow, in_v, in_v_pitch, cirow, cin_h, hipMemcpyHostToDevice, c->up));
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
            HIP_TRY(hipMemcpyAsync(out_y, d + off_oy, (off_ov - off_oy) + oc, hipMemcpyDeviceToHost, c->down));
        } else {
            HIP_TRY(copy_plane(out_y, out_y_pitch, d + off_oy, orow, orow, g.out_height, hipMemcpyDeviceToHost, c->down));
            if (chroma) {
                HIP_TRY(copy_plane(out_u, out_u_pitch, d + off_ou, crow, crow, cout_h, hipMemcpyDeviceToHost, c->down));
                HIP_TRY(copy_plane(out_v, out_v_pitch, d + off_ov, crow, crow, cout_h, hipMemcpyDeviceToHost, c->down));
            }
        }
        HIP_TRY(hipEventRecord(c->ev_done, c->down));
        c->done_pending = true;
        return RAISR_HIP_OK;
    }
    hipStream_t s = c->stream, s2 = c->stream2;
    // Y: upload, RAISR passes, download on the context stream; chroma (plain cheap upscale, Raisr.cpp:1373-1388)
    // runs on a second stream so its PCIe transfers overlap the Y kernels.
    const size_t irow = (size_t)g.in_width * bps, orow = (size_t)g.out_width * bps;
    const size_t cirow = (size_t)cin_w * bps, crow = (size_t)cout_w * bps;
    if (do_up) {
        HIP_TRY(copy_plane(d, irow, in_y, in_y_pitch, irow, g.in_height, hipMemcpyHostToDevice, s));
        if (c->blending == RAISR_HIP_BLEND_RANDOMNESS && y_keep > 0)   // pixels the reference leaves untouched keep the caller's bytes
            HIP_TRY(copy_plane(d + off_oy + y_skip * orow, orow, out_y, out_y_pitch, orow, y_keep, hipMemcpyHostToDevice, s));
        int rc = raisr_hip_process_y_device(c, d, irow, d + off_oy, orow, s);
        if (rc) return rc;
        if (chroma) {
            HIP_TRY(copy_plane(d + off_iu, cirow, in_u, in_u_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
            HIP_TRY(copy_plane(d + off_iv, cirow, in_v, in_v_pitch, cirow, cin_h, hipMemcpyHostToDevice, s2));
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
        const bool packed = chroma && !rows && out_y_pitch == orow && out_u_pitch == crow && out_v_pitch == crow &&
                            (const char*)out_u == (const char*)out_y + (off_ou - off_oy) && (const char*)out_v == (const char*)out_y + (off_ov - off_oy);
        if (packed) {
            if (!c->ev_chroma) HIP_TRY(hipEventCreateWithFlags(&c->ev_chroma, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->ev_chroma, s2));
            HIP_TRY(hipStreamWaitEvent(s, c->ev_chroma, 0));
            HIP_TRY(hipMemcpyAsync(out_y, d + off_oy, (off_ov - off_oy) + oc, hipMemcpyDeviceToHost, s));
        } else {
            if (chroma && c_keep > 0) {
                HIP_TRY(copy_plane(out_u, out_u_pitch, d + off_ou + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
                HIP_TRY(copy_plane(out_v, out_v_pitch, d + off_ov + c_skip * crow, crow, crow, c_keep, hipMemcpyDeviceToHost, s2));
            }
            if (y_keep > 0) HIP_TRY(copy_plane(out_y, out_y_pitch, d + off_oy + y_skip * orow, orow, orow, y_keep, hipMemcpyDeviceToHost, s));
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
    if (collect && !c->d_cert_stats) HIP_TRY(hipMalloc((void**)&c->d_cert_stats, 3 * sizeof(unsigned)));
    if (c->d_cert_stats) { HIP_TRY(hipMemsetAsync(c->d_cert_stats, 0, 3 * sizeof(unsigned), c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    if (!collect && c->d_cert_stats) { (void)hipFree(c->d_cert_stats); c->d_cert_stats = nullptr; }
    c->cert_check = check != 0;
    return RAISR_HIP_OK;
}

int raisr_hip_debug_certify_stats(raisr_hip_ctx* c, unsigned out[3])
{
    if (!c || !out) return fail(RAISR_HIP_EINVAL, "null argument");
    if (!c->d_cert_stats) return fail(RAISR_HIP_ESTATE, "statistics are not being collected");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_cert_stats, 3 * sizeof(unsigned), hipMemcpyDeviceToHost));
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

}  // extern "C"
