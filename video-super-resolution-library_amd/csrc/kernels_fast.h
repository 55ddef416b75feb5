// kernels_fast.h -- opt-in, NOT bit-exact fast mode: filter stage on the matrix cores (k_filter_mfma, k_build_mfma_bank)
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

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


