// kernels_experiments.h -- measured-and-rejected variants kept buildable for A/B runs (docs/EXPERIMENTS.md); included by
// device_abi.hip only in development builds (-DRAISR_HIP_DEV plus the RAISR_EXP_* macro of the experiment).  Not product code.
#pragma once

#ifdef RAISR_EXP_PERSIST
// Experiment: the same tile routine in a persistent grid (4 workgroups per CU walk the tiles in XCD-aware order).
#ifndef RAISR_EXP_PERSIST_WGS
#define RAISR_EXP_PERSIST_WGS 4
#endif
template <typename T>
__global__ __launch_bounds__(256, RAISR_EXP_PERSIST_WGS) void k_hashfilter_acp(const T* __restrict__ lr, PassParams P, GaussW gw, SepW S,
                                                           uint8_t* __restrict__ hash_out, float* __restrict__ hr, int tiles_x, int tiles_y,
                                                           int ncus, int skew_ticks, unsigned* __restrict__ tile_ctr)
{
    constexpr int TW = 64, TH = 16;
    constexpr int LW = 77, LH = TH + 12, GW_ = 74, GH = TH + 10;
    __shared__ float sL[LH * LW];
    using GT = typename GradOf<T>::type;
    __shared__ GT sG[GH * GW_];
    __shared__ typename FVec<4>::type sV[3 * 4 * GW_];
    uint2* sTab = reinterpret_cast<uint2*>(sV);
    __shared__ uint8_t sH[TH * TW];
    __shared__ uint8_t sH2[TH * TW];
    __shared__ uint16_t sList[kListMax];
    __shared__ unsigned sCnt[4];
    const unsigned ntiles = (unsigned)(tiles_x * tiles_y);
    if (skew_ticks > 0) {                                  // de-phase the workgroups that share a CU (k-th workgroup of a CU starts k * skew later)
        const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)((blockIdx.x / (unsigned)ncus) * (unsigned)skew_ticks);
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    __shared__ unsigned sTile;
    const unsigned xcd = blockIdx.x & 7u, n8 = ntiles & ~7u, per = n8 >> 3;
#pragma unroll 1
    for (unsigned t = blockIdx.x; ; t += gridDim.x) {
        int bx, by;
        if (tile_ctr) {                                    // dynamic: the next tile of this XCD's strip (one counter per XCD)
            if (threadIdx.x == 0) sTile = atomicAdd(&tile_ctr[xcd], 1u);
            __syncthreads();
            const unsigned j = sTile;
            unsigned u;
            if (j < per) u = xcd * per + j;
            else if (j == per && n8 + xcd < ntiles) u = n8 + xcd;
            else break;
            by = (int)(u / (unsigned)tiles_x); bx = (int)(u - (unsigned)by * (unsigned)tiles_x);
        } else {
            if (t >= ntiles) break;
            xcd_tile_of(t, (unsigned)tiles_x, ntiles, bx, by);
        }
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                      // opaque per tile: keeps the tile routine's lane-dependent set-up inside the loop
        __builtin_assume(tid < 256u);                      // (hoisted, it costs 52 spilled registers per lane)
        hashfilter_ac_tile<T, 0, LW, LH, GW_, GH, GT>(lr, P, gw, S, hash_out, hr, bx, by, sL, sG, sV, sTab, sH, sH2, sList, sCnt, tid);
#ifdef RAISR_EXP_PERSIST_LDSBAR
        lds_barrier();                                     // every wave is done with this tile's LDS (the HR stores stay in flight)
#else
        __syncthreads();                                   // every wave is done with this tile's LDS
#endif
    }
}
#endif
