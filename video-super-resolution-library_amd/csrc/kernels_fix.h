// kernels_fix.h -- k_fix_ac: the exact path of the pixels k_hashfilter_ac<.., DEFER> could not certify, as a small stream-ordered kernel
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_fix_ac.  The main kernel filtered every listed pixel with its APPROXIMATE bucket (hash_phase_defer); this kernel computes the
// reference's bucket of each (exact tensor in sumitup_ps_512's association, Raisr_AVX512.cpp:69-131; GetHashValue, :175-258) and,
// where it differs, redoes the pixel's filter step (DotProdPatch :134-149, accept test Raisr.cpp:1196-1200, tail re-hash rules as
// in filter_phase) and overwrites its HR value -- before k_blend reads the HR plane.  One wave per kFixTiles consecutive tiles, the
// entries of their 4 kFixTiles wave regions concatenated and taken 64 at a time in three phases:
//   1. exact tensors with 16 lanes per entry, four entries per round (exact_tensor16's association on the 13 x 13 LR window read straight
//      from the L2 / Infinity-Cache resident LR plane: 13 loads per lane, horizontal neighbours by DPP, the next round's loads in
//      flight), handed to lane (entry index mod 64) through the LDS crossbar (ds_bpermute);
//   2. the hash with ONE LANE PER ENTRY (the ~200-instruction hash costs the same for 1 or 64 active lanes);
//   3. the entries whose bucket changed (a minority: the approximate value is on the right side of the boundary more often than
//      not), four per round with 16 lanes each: eight window samples and eight coefficients per lane, the 16-lane chains and tree.
// LDS: the 1 KB table of VRCP14 / VRSQRT14 only, so that workgroups of this kernel fit next to the main kernel's on a busy CU.
// ------------------------------------------------------------------------------------------------
constexpr int kFixTiles = 4;          // tiles per wave of k_fix_ac (16 wave regions: ~60-100 entries on natural content, one or two full hash passes)

template <int CTRL>
__device__ __forceinline__ float row_shr_f(float v)      // DPP row_shr:n -- lane i reads lane i - n of its row of 16
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <typename T>
__global__ __launch_bounds__(256) void k_fix_ac(const T* __restrict__ lr, PassParams P, FixAc F, uint8_t* __restrict__ hash_out, float* __restrict__ hr,
                                                unsigned tile_first, unsigned tile_count)
{
    __shared__ uint2 sTab[128];
    if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, l = lane & 15;
    const unsigned t0 = (blockIdx.x * 4u + (unsigned)wv) * (unsigned)kFixTiles;          // first of the wave's tiles, within the launch's range
    if (t0 >= tile_count) return;
    const unsigned tile0 = tile_first + t0;                                               // ... its id within the frame
    const unsigned region0 = (blockIdx.z * F.zs_tiles + tile0) * 4u;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; hash_out += blockIdx.z * P.zs_hash;    // frame batches
    // the 16 region counts and their prefix sums (wave-uniform)
    unsigned pre[4 * kFixTiles + 1];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < kFixTiles; k++) {
        unsigned cw = (t0 + (unsigned)k < tile_count) ? *reinterpret_cast<const unsigned*>(F.counts + region0 + 4u * (unsigned)k) : 0u;
        cw = __builtin_amdgcn_readfirstlane(cw);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned n = (cw >> (8 * q)) & 0xFFu;
            if (n == 0xFFu) n = 0;                            // that wave ran the all-exact code itself
            pre[4 * k + q + 1] = pre[4 * k + q] + n;
        }
    }
    const unsigned total = pre[4 * kFixTiles];
    if (total == 0) return;
    // tile origins (the wave's tiles are consecutive in row-major order)
    int tr0[kFixTiles], tc0[kFixTiles];
    {
        int by = (int)(tile0 / (unsigned)F.tiles_x), bx = (int)(tile0 - (unsigned)by * (unsigned)F.tiles_x);
#pragma unroll
        for (int k = 0; k < kFixTiles; k++) {
            tr0[k] = kMargin + by * 16; tc0[k] = kMargin + bx * 64;
            if (++bx == F.tiles_x) { bx = 0; by++; }
        }
    }
    const uint16_t* list = F.entries + (size_t)region0 * kWaveCap;
    // entry e of the wave's tiles (regions concatenated): its pixel and approximate bucket
    auto fetch = [&](unsigned e, int& r, int& c, unsigned& bucket) {
        unsigned k = 0, base = 0;
#pragma unroll
        for (int i = 1; i < 4 * kFixTiles; i++) { const bool ge = e >= pre[i]; k += ge ? 1u : 0u; base = ge ? pre[i] : base; }
        const unsigned ent = list[k * kWaveCap + (e - base)];
        int rr = tr0[0], cc = tc0[0];
#pragma unroll
        for (int i = 1; i < kFixTiles; i++) { const bool hit = (k >> 2) == (unsigned)i; rr = hit ? tr0[i] : rr; cc = hit ? tc0[i] : cc; }
        r = rr + 4 * (int)(k & 3u) + (int)(ent & 3u);
        c = cc + (int)((ent >> 2) & 63u);
        bucket = ent >> 8;
    };
    // exact tensor with 16 lanes per entry: lane l <= 12 owns window column c - 6 + l (13 rows r-6 .. r+6: 13 loads); patch column
    // k = l - 1 lives in lanes 1..11 and takes its horizontal neighbours' samples from lanes l - 1 and l + 1 (DPP)
    const int lcol = min(l, 12);
    float wl[11];
#pragma unroll
    for (int i = 0; i < 11; i++) wl[i] = P.gauss_dev[min(max(l - 1, 0), 10) * 12 + i];
    const unsigned pitch = (unsigned)P.lr_pitch;

    for (unsigned e0 = 0; e0 < total; e0 += 64u) {
        const unsigned nb = min(64u, total - e0);
        int r = 0, c = 0;
        unsigned bucket = 0;
        fetch(e0 + min((unsigned)lane, nb - 1u), r, c, bucket);            // lane i <-> entry e0 + i (lanes past the batch repeat its last entry)
        // ---- phase 1: exact tensors, four entries per round, the next round's samples in flight during this round's arithmetic ----
        float A = 0.f, B = 0.f, D = 0.f;
        T Lraw[13];
        auto issue = [&](unsigned rd) {
            const int src = (int)min(4u * rd + (unsigned)g, nb - 1u);
            const int er = __shfl(r, src), ec = __shfl(c, src);
            const T* col = lr + (unsigned)(er - 6) * pitch + (unsigned)(ec - 6 + lcol);
#pragma unroll
            for (int j = 0; j < 13; j++) Lraw[j] = col[(unsigned)j * pitch];
        };
        issue(0);
        for (unsigned rd = 0; 4u * rd < nb; rd++) {
            float Lc[13];
#pragma unroll
            for (int j = 0; j < 13; j++) Lc[j] = (float)Lraw[j];
            if (4u * (rd + 1u) < nb) issue(rd + 1u);
            f2 AD = {0.f, 0.f};
            float Bs = 0.f;
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const float right = row_shl<0x101>(Lc[i + 1]), left = row_shr_f<0x111>(Lc[i + 1]);
                const f2 gg = {Lc[i + 2] - Lc[i], right - left};          // GetGx: row below - row above; GetGy: right - left
                const f2 w2 = {wl[i], wl[i]};
                const f2 pq = gg * w2;
                AD = __builtin_elementwise_fma(pq, gg, AD);
                Bs = __builtin_fmaf(pq.x, gg.y, Bs);
            }
            // patch column k sits in lane k + 1: move the column sums down one lane, then fold as sumitup_ps_512 does
            const bool lane3 = l == 3;
            const float a = fold11(row_shl<0x101>(AD.x), lane3), b = fold11(row_shl<0x101>(Bs), lane3), d = fold11(row_shl<0x101>(AD.y), lane3);   // valid in lane 0 of every group
            // lane i in [4 rd, 4 rd + 4) takes the result of group i & 3
            const int srcl = (lane & 3) * 16;
            const float av = __shfl(a, srcl), bv = __shfl(b, srcl), dv = __shfl(d, srcl);
            if ((unsigned)(lane >> 2) == rd) { A = av; B = bv; D = dv; }
        }
        // ---- phase 2: the hash, one lane per entry ----
        unsigned hA = 0xFFu, hB = 0xFFu;
        bool redo = false;
        if ((unsigned)lane < nb) {
            flavour_hash(P, sTab, A, B, D, c, hA, hB);
            redo = hA != bucket || (hB != 0xFFu && hB != bucket);
            if (P.write_hash) hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = (uint8_t)hA;
        }
        // ---- phase 3: the filter step of the entries whose bucket changed, 16 lanes per entry ----
        unsigned long long m = __ballot(redo);
        const unsigned hab = hA | (hB << 8);
        while (m) {
            int src[4];
            bool have[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                have[k] = m != 0;
                src[k] = have[k] ? (int)__builtin_ctzll(m) : 0;
                if (have[k]) m &= m - 1;
            }
            const int sl = g == 0 ? src[0] : (g == 1 ? src[1] : (g == 2 ? src[2] : src[3]));
            const bool on = g == 0 ? have[0] : (g == 1 ? have[1] : (g == 2 ? have[2] : have[3]));
            const int er = __shfl(r, sl), ec = __shfl(c, sl);
            const unsigned eh = __shfl(hab, sl);
            const unsigned hA2 = eh & 0xFFu, hB2 = (eh >> 8) & 0xFFu;
            // the lane's eight taps 16 ch + l of the 11 x 11 patch (padding taps k >= 121 carry coefficient +0: any finite sample will do)
            const T* win = lr + (unsigned)(er - 5) * pitch + (unsigned)(ec - 5);
            float x[8];
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const int k = 16 * ch + l;
                const int kk = k < kTaps ? k : 0;
                x[ch] = (float)win[(unsigned)(kk / 11) * pitch + (unsigned)(kk % 11)];
            }
            const unsigned type = (P.pixel_types == 4) ? (unsigned)(((er - 5) & 1) * 2 + ((ec - 5) & 1)) : 0u;
            auto dot = [&](unsigned h) -> float {                 // the lane's eight coefficients: two 16-byte loads from the lane-major bank
                const float4* f = reinterpret_cast<const float4*>(P.bank_lm + ((size_t)h * (unsigned)P.pixel_types + type) * kLmRow + 4 * l);
                const float4 fa = f[0], fb = f[16];
                float acc = x[0] * fa.x;
                acc = __builtin_fmaf(x[1], fa.y, acc); acc = __builtin_fmaf(x[2], fa.z, acc); acc = __builtin_fmaf(x[3], fa.w, acc);
                acc = __builtin_fmaf(x[4], fb.x, acc); acc = __builtin_fmaf(x[5], fb.y, acc); acc = __builtin_fmaf(x[6], fb.z, acc); acc = __builtin_fmaf(x[7], fb.w, acc);
                return tree16(acc);
            };
            const float centre = (float)win[5u * pitch + 5u];
            float keep = centre;
            if (on) {
                const float vA = dot(hA2);
                if (vA > P.lo && vA < P.hi) keep = vA;
                if (hB2 != 0xFFu) {                             // tail column: AVX2 re-hash (keep-first-if-rejected; Randomness blends the last candidate)
                    const float vB = dot(hB2);
                    if (vB > P.lo && vB < P.hi) keep = vB;
                    else if (P.randomness) keep = centre;
                }
                if (l == 0) hr[(size_t)er * P.hr_pitch + ec] = keep;
            }
        }
    }
}
