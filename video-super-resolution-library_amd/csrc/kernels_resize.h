// kernels_resize.h -- cheap upscale (stand-in for ippiResizeLinear, Raisr.cpp:947-958): k_resize, k_resize2x, k_copy
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_resize: centre-aligned bilinear with replicate border in exact integer arithmetic.
// dst(y,x): n = (2d+1)*S - D, den = 2D per axis (reduced by gcd on the host: Sx,Dx,Sy,Dy).
// ------------------------------------------------------------------------------------------------
struct ResizeParams {
    int sw, sh, dw, dh;
    int spitch, dpitch;          // in elements
    int Sx, Dx, Sy, Dy;          // reduced ratios
    int tie_even;
    int sstep, dstep;            // samples between horizontally adjacent samples of the plane (1; 2 = one channel of an interleaved
                                 // two-channel plane: the UV plane of NV12 / P010); pitches stay in samples of the container
    int narrow;                  // 1: the 32-bit path of k_resize is exact for this geometry (make_resize)
    float rdenx, rdeny, rden2;   // 1 / (2 Dx), 1 / (2 Dy), 1 / (2 * 2 Dx * 2 Dy)
    size_t zs_src, zs_dst;       // frame batches: blockIdx.z = frame, planes zs_* elements apart (0 for a single plane)
    int in_shift, out_shift;     // MSB-aligned samples of device frames (P010: value << 6, VideoDataType::bitShift): sample = stored >> in_shift,
                                 // stored = sample << out_shift (the reference's OpenCL pre/post-process kernels, Raisr_OpenCL_kernel.h:241-276)
};

__device__ __forceinline__ void axis_tap(int d, int S, int D, int size, int& i0, int& i1, int& f)
{
    const int n = (2 * d + 1) * S - D, den = 2 * D;
    int q = n >= 0 ? n / den : -((-n + den - 1) / den);
    f = n - q * den;
    i0 = min(max(q, 0), size - 1);
    i1 = min(max(q + 1, 0), size - 1);
}

// floor(x / d) for 0 <= x < 2^31 and a kernel-uniform divisor 0 < d < 2^16 with rd = 1.0f / d from the host: the fp32 estimate
// is within one of the quotient (relative error < 2^-22, quotient < 2^31 / d), two compare-and-adjust steps make it exact.
// (The generic 32-bit division expands to ~30 instructions, the 64-bit one of the wide path below to ~150.)
__device__ __forceinline__ unsigned div_small(unsigned x, unsigned d, float rd)
{
    unsigned q = (unsigned)((float)x * rd);
    int r = (int)(x - q * d);
    if (r < 0) { q--; r += (int)d; }
    if (r < 0) { q--; r += (int)d; }
    if (r >= (int)d) { q++; r -= (int)d; }
    if (r >= (int)d) q++;
    return q;
}

// axis_tap with n + den >= 0 folded in (d >= 0, S <= 2 D: any upscale): floor(n / den) = floor((n + den) / den) - 1
__device__ __forceinline__ void axis_tap_small(int d, int S, int D, float rden, int size, int& i0, int& i1, int& f)
{
    const unsigned den = 2u * (unsigned)D;
    const unsigned n1 = (unsigned)((2 * d + 1) * S - D + (int)den);
    const unsigned q1 = div_small(n1, den, rden);
    f = (int)(n1 - q1 * den);
    const int q = (int)q1 - 1;
    i0 = min(max(q, 0), size - 1);
    i1 = min(max(q + 1, 0), size - 1);
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_resize(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    src += blockIdx.z * R.zs_src; dst += blockIdx.z * R.zs_dst;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= R.dw || y >= R.dh) return;
    if (R.narrow) {
        // every intermediate fits 32 bits (checked on the host: 2 denx deny 65535 + denx deny < 2^31, S <= 2 D, coordinates < 2^15)
        int x0, x1, fx, y0, y1, fy;
        axis_tap_small(x, R.Sx, R.Dx, R.rdenx, R.sw, x0, x1, fx);
        axis_tap_small(y, R.Sy, R.Dy, R.rdeny, R.sh, y0, y1, fy);
        const unsigned denx = 2u * (unsigned)R.Dx, deny = 2u * (unsigned)R.Dy;
        const TIn* r0 = src + (size_t)y0 * R.spitch;
        const TIn* r1 = src + (size_t)y1 * R.spitch;
        x0 *= R.sstep; x1 *= R.sstep;
        const int sh = R.in_shift;
        const unsigned top = (denx - (unsigned)fx) * ((unsigned)r0[x0] >> sh) + (unsigned)fx * ((unsigned)r0[x1] >> sh);
        const unsigned bot = (denx - (unsigned)fx) * ((unsigned)r1[x0] >> sh) + (unsigned)fx * ((unsigned)r1[x1] >> sh);
        const unsigned num = (deny - (unsigned)fy) * top + (unsigned)fy * bot;
        const unsigned den = denx * deny;
        const unsigned t = 2u * num + den;
        unsigned q = div_small(t, 2u * den, R.rden2);
        if (R.tie_even && (t - q * 2u * den == 0u) && (q & 1u)) q--;
        dst[(size_t)y * R.dpitch + (size_t)x * R.dstep] = (TOut)(q << R.out_shift);
        return;
    }
    int x0, x1, fx, y0, y1, fy;
    axis_tap(x, R.Sx, R.Dx, R.sw, x0, x1, fx);
    axis_tap(y, R.Sy, R.Dy, R.sh, y0, y1, fy);
    const long long denx = 2 * R.Dx, deny = 2 * R.Dy;
    const TIn* r0 = src + (size_t)y0 * R.spitch;
    const TIn* r1 = src + (size_t)y1 * R.spitch;
    x0 *= R.sstep; x1 *= R.sstep;
    const int sh = R.in_shift;
    const long long top = (denx - fx) * (long long)(r0[x0] >> sh) + (long long)fx * (r0[x1] >> sh);
    const long long bot = (denx - fx) * (long long)(r1[x0] >> sh) + (long long)fx * (r1[x1] >> sh);
    const long long num = (deny - fy) * top + fy * bot;
    const long long den = denx * deny;
    long long q = (2 * num + den) / (2 * den);
    if (R.tie_even && ((2 * num + den) % (2 * den) == 0) && (q & 1)) q--;
    dst[(size_t)y * R.dpitch + (size_t)x * R.dstep] = (TOut)(q << R.out_shift);
}

// 2x special case of the same arithmetic: weights {1,3}/4 per axis, out = (sum + 8) >> 4
// (ties: +8 then >>4 is round-half-up; the half-even switch subtracts one when the discarded
// bits are exactly 8 and the quotient is odd).  One thread produces an 8 x 2 block of output pixels -- output rows 2 sy, 2 sy + 1 and
// columns 8 t .. 8 t + 7 -- from source rows sy - 1, sy, sy + 1 and source columns 4 t - 1 .. 4 t + 4 (replicate-clamped): the
// vertical sums 3 b + a / 3 b + c share 3 b, the horizontal ones share 3 v.  Round 5: the 4 x 1 version spent 29 vector
// instructions per output pixel (3.8 M per 4K frame, 4 % of the pipeline's), this one 7; the kernel runs beside the other frames'
// issue-bound k_hashfilter_ac, so its instructions are what it costs (docs/EXPERIMENTS.md R5.9).
template <typename TIn>
__device__ __forceinline__ void resize2x_row6(const TIn* __restrict__ src, unsigned row_off, const unsigned (&col)[6], int sh, int (&v)[6])
{
    // six single-sample loads at 32-bit offsets from the plane's (uniform) base: fewer vector instructions than one wide load plus
    // its unpacking, and no branch for the planes' edges (planes are < 2^31 samples)
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = (int)src[row_off + col[k]];
    if (sh) {
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] >>= sh;
    }
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_resize2x(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    src += blockIdx.z * R.zs_src; dst += blockIdx.z * R.zs_dst;
    int bx, by;
    xcd_tile(bx, by);                                         // vertically adjacent blocks share input rows: keep them on one XCD
    const int t = bx * 64 + (threadIdx.x & 63);               // group of 8 output columns
    const int sy = by * 4 + (threadIdx.x >> 6);               // source row = pair of output rows
    const int x0 = 8 * t, c = 4 * t;
    if (x0 >= R.dw || sy >= R.sh) return;
    int a[6], b[6], d[6];
    unsigned col[6];                                          // source columns c - 1 .. c + 4, replicate-clamped (c itself is inside: 8 t < 2 sw)
    col[0] = (unsigned)max(c - 1, 0); col[1] = (unsigned)c;
#pragma unroll
    for (int k = 2; k < 6; k++) col[k] = (unsigned)min(c - 1 + k, R.sw - 1);
    resize2x_row6(src, (unsigned)max(sy - 1, 0) * (unsigned)R.spitch, col, R.in_shift, a);
    resize2x_row6(src, (unsigned)sy * (unsigned)R.spitch, col, R.in_shift, b);
    resize2x_row6(src, (unsigned)min(sy + 1, R.sh - 1) * (unsigned)R.spitch, col, R.in_shift, d);
    int o[2][8];
    {
        int vt[6], vb[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int b3 = 3 * b[k];
            vt[k] = b3 + a[k];                                // output row 2 sy:     (sy - 1: 1, sy: 3)
            vb[k] = b3 + d[k];                                // output row 2 sy + 1: (sy: 3, sy + 1: 1)
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t3 = 3 * vt[k + 1], u3 = 3 * vb[k + 1];
            o[0][2 * k] = vt[k] + t3;     o[0][2 * k + 1] = t3 + vt[k + 2];
            o[1][2 * k] = vb[k] + u3;     o[1][2 * k + 1] = u3 + vb[k + 2];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = 2 * sy + r;
        if (y >= R.dh) break;
        if (R.tie_even) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int q = (o[r][i] + 8) >> 4;
                if (((o[r][i] & 15) == 8) && (q & 1)) q--;
                o[r][i] = q << R.out_shift;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) o[r][i] = ((o[r][i] + 8) >> 4) << R.out_shift;
        }
        TOut* p = dst + (unsigned)y * (unsigned)R.dpitch + (unsigned)x0;
        if (x0 + 7 < R.dw && (reinterpret_cast<uintptr_t>(p) & (8 * sizeof(TOut) - 1)) == 0) {
            if (sizeof(TOut) == 1) {
                *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)o[r][0] | ((unsigned)o[r][1] << 8) | ((unsigned)o[r][2] << 16) | ((unsigned)o[r][3] << 24),
                                                          (unsigned)o[r][4] | ((unsigned)o[r][5] << 8) | ((unsigned)o[r][6] << 16) | ((unsigned)o[r][7] << 24));
            } else {
                *reinterpret_cast<uint4*>(p) = make_uint4((unsigned)o[r][0] | ((unsigned)o[r][1] << 16), (unsigned)o[r][2] | ((unsigned)o[r][3] << 16),
                                                          (unsigned)o[r][4] | ((unsigned)o[r][5] << 16), (unsigned)o[r][6] | ((unsigned)o[r][7] << 16));
            }
        } else {
            for (int i = 0; i < 8 && x0 + i < R.dw; i++) p[i] = (TOut)o[r][i];
        }
    }
}

// 3:2 special case (the reference's 1.5x models): n = 4 d - 1 over den = 6 per axis, so output d = 3 m + e reads sources
// 2 m - 1 + e and 2 m + e with weights {1,5}, {3,3}, {5,1} (/6) for e = 0, 1, 2; out = floor((2 num + 36) / 72) [half-up]
// = floor((num + 18) / 36), num <= 36 * 65535.  Same arithmetic as k_resize.
// One thread produces a 12 x 3 block of output pixels -- output rows 3 n .. 3 n + 2 and columns 12 j .. 12 j + 11 -- from source rows
// 2 n - 1 .. 2 n + 2 and source columns 8 j - 1 .. 8 j + 8 (replicate-clamped): 40 single-sample loads, 4 vertical sums per source
// column, one or two operations, an addition and ONE multiply-high per output (floor(x / 36) = mulhi(x, ceil(2^32 / 36)) for
// x < 2^27), three 12-sample stores.  Round 6: the 3 x 1 version spent 38 vector and 3.7 memory instructions per output pixel
// (5.5 % of C4's GPU time); the kernel runs beside the other frames' issue-bound k_hashfilter16, so its instructions are what it
// costs (docs/EXPERIMENTS.md R5.9, R6.10).
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_resize3x2(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    src += blockIdx.z * R.zs_src; dst += blockIdx.z * R.zs_dst;
    int bx, by;
    xcd_tile(bx, by);                                         // vertically adjacent blocks share input rows: keep them on one XCD
    const int j = bx * 64 + (threadIdx.x & 63);               // group of 12 output columns
    const int n = by * 4 + (threadIdx.x >> 6);                // group of 3 output rows
    const int x0 = 12 * j, y0 = 3 * n;
    if (x0 >= R.dw || y0 >= R.dh) return;
    unsigned col[10], row[4];                                 // source columns 8 j - 1 .. 8 j + 8, rows 2 n - 1 .. 2 n + 2, replicate-clamped
#pragma unroll
    for (int k = 0; k < 10; k++) col[k] = (unsigned)min(max(8 * j - 1 + k, 0), R.sw - 1);
#pragma unroll
    for (int k = 0; k < 4; k++) row[k] = (unsigned)min(max(2 * n - 1 + k, 0), R.sh - 1) * (unsigned)R.spitch;
    unsigned s[4][10];                                        // single-sample loads at 32-bit offsets (planes are < 2^31 samples), as k_resize2x
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 10; k++) s[r][k] = (unsigned)src[row[r] + col[k]];
    if (R.in_shift) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 10; k++) s[r][k] >>= R.in_shift;
    }
    const bool vec = (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && (((unsigned)R.dpitch * sizeof(TOut)) & 3) == 0 && x0 + 11 < R.dw;   // x0 * sizeof(TOut) is a multiple of 4
#pragma unroll
    for (int ey = 0; ey < 3; ey++) {
        const int y = y0 + ey;
        if (y >= R.dh) break;
        // output row 3 n + ey: source rows (2 n - 1 + ey, 2 n + ey) = s[ey], s[ey + 1] with weights (gy, fy) = (1,5), (3,3), (5,1)
        unsigned v[10];
#pragma unroll
        for (int k = 0; k < 10; k++)
            v[k] = ey == 0 ? 5u * s[1][k] + s[0][k] : (ey == 1 ? 3u * (s[1][k] + s[2][k]) : 5u * s[2][k] + s[3][k]);
        unsigned o[12];
#pragma unroll
        for (int g = 0; g < 4; g++) {                         // outputs 3 g + e of the block: group m = 4 j + g reads v[2 g + e], v[2 g + e + 1]
            o[3 * g + 0] = 5u * v[2 * g + 1] + v[2 * g];             // weights (1, 5)
            o[3 * g + 1] = 3u * (v[2 * g + 1] + v[2 * g + 2]);       //         (3, 3)
            o[3 * g + 2] = 5u * v[2 * g + 2] + v[2 * g + 3];         //         (5, 1)
        }
        // (2 num + 36) / 72 = (num + 18) / 36;  floor(t / 36) = mulhi(t, 119304648): 119304648 = ceil(2^32 / 36), exact for t < 2^27
        if (R.tie_even) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const unsigned t = o[i] + 18u;
                unsigned q = __umulhi(t, 119304648u);
                if (t - 36u * q == 0u && (q & 1u)) q--;
                o[i] = q;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; i++) o[i] = __umulhi(o[i] + 18u, 119304648u);
        }
        if (R.out_shift) {
#pragma unroll
            for (int i = 0; i < 12; i++) o[i] <<= R.out_shift;
        }
        TOut* p = dst + (size_t)y * R.dpitch + x0;
        if (vec) {
            typedef unsigned u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));      // the block starts on a 4-byte, not a 16-byte boundary: says so
            if (sizeof(TOut) == 1) {
                u32x3a4 w;
#pragma unroll
                for (int k = 0; k < 3; k++) w[k] = (o[4 * k] & 0xffu) | ((o[4 * k + 1] & 0xffu) << 8) | ((o[4 * k + 2] & 0xffu) << 16) | (o[4 * k + 3] << 24);
                *reinterpret_cast<u32x3a4*>(p) = w;
            } else {
                u32x3a4 w0, w1;
#pragma unroll
                for (int k = 0; k < 3; k++) { w0[k] = (o[2 * k] & 0xffffu) | (o[2 * k + 1] << 16); w1[k] = (o[6 + 2 * k] & 0xffffu) | (o[7 + 2 * k] << 16); }
                *reinterpret_cast<u32x3a4*>(p) = w0;                        // (a 3-vector occupies 16 bytes as an array element: address the second half by hand)
                *reinterpret_cast<u32x3a4*>(p + 6) = w1;
            }
        } else {
            for (int i = 0; i < 12 && x0 + i < R.dw; i++) p[i] = (TOut)o[i];
        }
    }
}

// same-size case of the cheap upscale (two-pass mode 2 runs pass 1 at input size, Raisr.cpp:960-975): widen/copy
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_copy(const TIn* __restrict__ src, TOut* __restrict__ dst, ResizeParams R)
{
    src += blockIdx.z * R.zs_src; dst += blockIdx.z * R.zs_dst;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < R.dw && y < R.dh) dst[(size_t)y * R.dpitch + x] = (TOut)((src[(size_t)y * R.spitch + x] >> R.in_shift) << R.out_shift);
}

