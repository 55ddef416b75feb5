// kernels_blend.h -- census-transform blend and border policy: k_blend, k_blend_rand
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

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
    by += P.tile_y0;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; out += blockIdx.z * P.zs_out;          // frame batches
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
        out[(size_t)y * out_pitch + x] = (TOut)(iv << P.out_shift);
    }
}

// ------------------------------------------------------------------------------------------------
// k_blend4 / k_blend4_16: the same two stages (fp32: CTCountOfBitsChangedSegment_AVX256_32f, Raisr_AVX256.cpp:68-166; binary16:
// CTCountOfBitsChangedSegment_AVX512FP16_16f, Raisr_AVX512FP16.cpp:258-355; border policy Raisr.cpp:999-1028,1252-1265) for planes
// whose rows are 4-sample aligned (W % 4 == 0, aligned bases: every video size) -- the production kernels; k_blend / k_blend16 above
// stay as the path for every other geometry.  No LDS, no barrier: a lane owns FOUR adjacent columns, a wave 256 columns x RW rows.
// A row of a plane is one aligned load per lane (4 samples: 4 / 8 bytes of LR, 16 / 8 bytes of HR) plus one one-sample load for
// the two columns outside the wave (lane 63: column c0 + 256, every other lane: column c0 - 1); the left / right neighbours of a
// lane's outer columns come from the adjacent lanes through DPP wave shifts (wave_shr:1 / wave_shl:1, whose `old` operand is
// exactly that outside column for lane 0 / lane 63).  The 3-row window slides down in registers, rows are requested one ahead of
// their use.  Arithmetic per pixel: the expressions of k_blend / k_blend16 (same operations, same order); the fp32 variant's final
// integer clamp max(min(cvttps(floor v), hi), lo) is evaluated on the float (blend4_row: same integer for every input).
// ------------------------------------------------------------------------------------------------
template <typename TOut> struct Lr4;
template <> struct Lr4<uint8_t> {
    typedef uint32_t raw;
    static __device__ __forceinline__ void unpack(raw v, float* f) {
        f[0] = (float)(v & 0xffu); f[1] = (float)((v >> 8) & 0xffu); f[2] = (float)((v >> 16) & 0xffu); f[3] = (float)(v >> 24);
    }
    static __device__ __forceinline__ raw pack(const int* iv) {
        return ((uint32_t)iv[0] & 0xffu) | (((uint32_t)iv[1] & 0xffu) << 8) | (((uint32_t)iv[2] & 0xffu) << 16) | ((uint32_t)iv[3] << 24);
    }
};
template <> struct Lr4<uint16_t> {
    typedef uint2 raw;
    static __device__ __forceinline__ void unpack(raw v, float* f) {
        f[0] = (float)(v.x & 0xffffu); f[1] = (float)(v.x >> 16); f[2] = (float)(v.y & 0xffffu); f[3] = (float)(v.y >> 16);
    }
    static __device__ __forceinline__ raw pack(const int* iv) {
        return make_uint2(((uint32_t)iv[0] & 0xffffu) | ((uint32_t)iv[1] << 16), ((uint32_t)iv[2] & 0xffffu) | ((uint32_t)iv[3] << 16));
    }
};
template <bool HR16> struct Hr4;
template <> struct Hr4<false> {
    typedef float4 raw; typedef float one;
    static __device__ __forceinline__ void unpack(raw v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    static __device__ __forceinline__ float cvt(one v) { return v; }
};
template <> struct Hr4<true> {
    typedef uint2 raw; typedef uint16_t one;
    static __device__ __forceinline__ void unpack(raw v, float* f) {
        f[0] = (float)h_bits((uint16_t)(v.x & 0xffffu)); f[1] = (float)h_bits((uint16_t)(v.x >> 16));
        f[2] = (float)h_bits((uint16_t)(v.y & 0xffffu)); f[3] = (float)h_bits((uint16_t)(v.y >> 16));
    }
    static __device__ __forceinline__ float cvt(one v) { return (float)h_bits(v); }
};

// lane i <- v of lane i - 1 (lane 0 keeps `edge`) / lane i <- v of lane i + 1 (lane 63 keeps `edge`)
__device__ __forceinline__ float from_left_lane(float edge, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right_lane(float edge, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// one output row of a lane: four pixels from the 3 x 6 windows l (LR) / h (HR); ALL16: every column of this wave is in the binary16 body
template <typename TOut, bool HR16, bool ALL16>
__device__ __forceinline__ void blend4_row(const float (&l)[3][6], const float (&h)[3][6], const PassParams& P, int c_avx, int x0, int* iv)
{
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float Lc = l[1][p + 1], Hc = h[1][p + 1];
        int hd = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (i == 1 && j == 1) continue;
                hd += ((l[i][p + j] < Lc) != (h[i][p + j] < Hc));
            }
        auto tail32 = [&]() {                                      // k_blend's fp32 expressions
            const float weight = (float)hd * 0.125f;               // hd / 8.0f exactly
            const float w2 = 1.0f - weight;
            float val = (weight * Lc) + (w2 * Hc);
            val = val + 0.5f;
            return val;
        };
        if (!HR16) {
            // k_blend: iv = cvttps(floor(val)) (0x80000000 when out of range or NaN), then max(min(iv, ihi), ilo).  floor(val) is an
            // integer and so are the limits: clamping it as a float first and converting then gives the same integer; what cvttps turns
            // into INT_MIN (>= 2^31, NaN -- the comparison is false for both) is sent to the lower limit, where INT_MIN ends up.
            const float fl = __builtin_floorf(tail32());
            const float g = fl < 2147483648.0f ? fl : P.lo;
            iv[p] = (int)__builtin_amdgcn_fmed3f(g, P.lo, P.hi);
        } else {
            auto body16 = [&]() {                                  // 32-wide binary16 body (:303-312)
                const hf weight = (hf)((float)hd * 0.125f);        // hd / 8 exactly
                const hf w2 = (hf)1.0f - weight;
                hf val = (weight * (hf)Lc) + (w2 * (hf)Hc);
                val = val + (hf)0.5f;
                const float fl = __builtin_floorf((float)val);
                int fi = (fl >= -32768.0f && fl <= 32767.0f) ? (int)fl : -32768;
                if (fi < 0) fi = 0xFFFF;                            // cvtph_epu16 of a negative value
                return max(min(fi, P.ihi), P.ilo);
            };
            if (ALL16) iv[p] = body16();
            else if (x0 + p < c_avx) iv[p] = body16();
            else {                                                 // scalar fp32 tail (:326-352)
                const float val = tail32();
                const float cl = val < P.lo ? P.lo : (val > P.hi ? P.hi : val);
                iv[p] = (int)cl;
            }
        }
    }
}

template <typename TOut, bool HR16, int RW>
__device__ __forceinline__ void blend4_wave(const TOut* __restrict__ lr, const typename Hr4<HR16>::one* __restrict__ hr, const PassParams& P,
                                            int c_avx, TOut* __restrict__ out, int out_pitch)
{
    static_assert(RW == 4 || RW == 8 || RW == 16, "rows of a wave");
    constexpr int NRG = 16 / RW, NCG = 4 / NRG;             // a workgroup: NRG row groups x NCG column groups of one wave each = (256 NCG) x 16 pixels
    typedef typename Lr4<TOut>::raw LRaw;
    typedef typename Hr4<HR16>::raw HRaw;
    typedef typename Hr4<HR16>::one HOne;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, by;
    xcd_tile(bx, by);
    by += P.tile_y0;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; out += blockIdx.z * P.zs_out;          // frame batches
    const int c0 = (bx * NCG + w % NCG) * 256, r0 = by * 16 + (w / NCG) * RW;
    if (c0 >= P.W || r0 >= P.H) return;                      // whole waves only: every lane of a running wave stays active (DPP sources)
    const int x0 = c0 + 4 * lane;
    const int xl = min(x0, P.W - 4);                         // lanes right of the plane repeat its last four columns and store nothing
    const int xh = lane == 63 ? min(c0 + 256, P.W - 1) : max(c0 - 1, 0);
    bool zc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) zc[k] = xl + k >= kMargin && xl + k < P.c_final;
    const bool zh = xh >= kMargin && xh < P.c_final;
    const bool all16 = HR16 && c0 + 256 <= c_avx;            // wave-uniform

    // Every row of both planes is read, rows outside the filtered zone too (the HR plane is allocated whole; what it holds there is
    // never used: HR := LR outside the zone, Raisr.cpp:1035) -- no branch between the loads.
    struct Raw { LRaw l4; TOut l1; HRaw h4; HOne h1; };
    auto request = [&](int i, Raw& R) {                      // window row i of the wave = plane row r0 - 1 + i, replicate-clamped
        const int gy = min(max(r0 - 1 + i, 0), P.H - 1);
        const TOut* lrow = lr + (size_t)gy * P.lr_pitch;
        const HOne* hrow = hr + (size_t)gy * P.hr_pitch;
        R.l4 = *(const LRaw*)(lrow + xl);
        R.h4 = *(const HRaw*)(hrow + xl);
        R.l1 = lrow[xh];
        R.h1 = hrow[xh];
    };
    auto unpack = [&](int i, const Raw& R, float* l, float* h) {    // l[0..5], h[0..5]: columns x0 - 1 .. x0 + 4
        const int gy = min(max(r0 - 1 + i, 0), P.H - 1);
        const bool rz = gy >= kMargin && gy < P.H - kMargin;       // wave-uniform
        Lr4<TOut>::unpack(R.l4, l + 1);
        const float le = (float)R.l1;
        l[0] = from_left_lane(le, l[4]);
        l[5] = from_right_lane(le, l[1]);
        float hv[4];
        Hr4<HR16>::unpack(R.h4, hv);
#pragma unroll
        for (int k = 0; k < 4; k++) h[1 + k] = (rz && zc[k]) ? hv[k] : l[1 + k];
        const float he = (rz && zh) ? Hr4<HR16>::cvt(R.h1) : le;
        h[0] = from_left_lane(he, h[4]);
        h[5] = from_right_lane(he, h[1]);
    };

    Raw q[RW + 2];
    // Three rows are requested before the first is consumed, one more per output row: enough to cover the latency beside the other
    // frames' main kernel, and 56-65 registers -- a wave of this kernel fits into what four waves of k_hashfilter_ac leave of a
    // SIMD's 512 (96 beside the symmetric stage's 4 x 104, 64 beside the eight-load stage's 4 x 112).  Requesting six rows first and
    // four at a time (94-113 registers) was no faster anywhere and 0.7 % slower on the 16-bit planes of C5 (R6.9).
#pragma unroll
    for (int i = 0; i < 3; i++) request(i, q[i]);
    float l[3][6], h[3][6];
    unpack(0, q[0], l[1], h[1]);
    unpack(1, q[1], l[2], h[2]);
#pragma unroll
    for (int rr = 0; rr < RW; rr++) {
        if (rr + 3 < RW + 2) request(rr + 3, q[rr + 3]);
        const int y = r0 + rr;                               // rows below the plane are computed on repeated rows and not stored
#pragma unroll
        for (int j = 0; j < 6; j++) { l[0][j] = l[1][j]; l[1][j] = l[2][j]; h[0][j] = h[1][j]; h[1][j] = h[2][j]; }
        unpack(rr + 2, q[rr + 2], l[2], h[2]);
        int iv[4];
        if (all16) blend4_row<TOut, HR16, true>(l, h, P, c_avx, x0, iv);
        else blend4_row<TOut, HR16, false>(l, h, P, c_avx, x0, iv);
        // border policy: row 0, row H - 1, column 0, column W - 1 keep the unclamped LR value (W % 4 == 0: column W - 1 is a lane's fourth)
        const bool yb = y == 0 || y == P.H - 1;              // wave-uniform
        iv[0] = (yb || x0 == 0) ? (int)l[1][1] : iv[0];
        iv[1] = yb ? (int)l[1][2] : iv[1];
        iv[2] = yb ? (int)l[1][3] : iv[2];
        iv[3] = (yb || x0 + 3 == P.W - 1) ? (int)l[1][4] : iv[3];
#pragma unroll
        for (int p = 0; p < 4; p++) iv[p] = iv[p] << P.out_shift;
        if (x0 < P.W && y < P.H) *(LRaw*)(out + (size_t)y * out_pitch + x0) = Lr4<TOut>::pack(iv);
    }
}

template <typename TOut, int RW>
__global__ __launch_bounds__(256) void k_blend4(const TOut* __restrict__ lr, const float* __restrict__ hr, PassParams P, TOut* __restrict__ out, int out_pitch)
{
    blend4_wave<TOut, false, RW>(lr, hr, P, 0, out, out_pitch);
}
template <typename TOut, int RW>
__global__ __launch_bounds__(256) void k_blend4_16(const TOut* __restrict__ lr, const uint16_t* __restrict__ hr, PassParams P, int c_avx, TOut* __restrict__ out, int out_pitch)
{
    blend4_wave<TOut, true, RW>(lr, hr, P, c_avx, out, out_pitch);
}

// rows of 4-sample aligned planes: the geometry k_blend4 / k_blend4_16 need
template <typename TOut>
inline bool blend4_fits(const void* lr, const void* hr, const void* out, int out_pitch, const PassParams& P, size_t hr_elem)
{
    const size_t a = 4 * sizeof(TOut), ah = 4 * hr_elem;
    return P.W >= 8 && P.W % 4 == 0 && P.lr_pitch % 4 == 0 && P.hr_pitch % 4 == 0 && out_pitch % 4 == 0
        && (uintptr_t)lr % a == 0 && (uintptr_t)out % a == 0 && (uintptr_t)hr % ah == 0
        && P.zs_lr % 4 == 0 && P.zs_hr % 4 == 0 && P.zs_out % 4 == 0;
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
        if (!untouched) out[(size_t)y * out_pitch + x] = (TOut)((int)lc << P.out_shift);
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
    out[(size_t)y * out_pitch + x] = (TOut)((int)cl << P.out_shift);
}


