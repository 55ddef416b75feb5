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


