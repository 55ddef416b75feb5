// kernels_fix.h -- k_fix_ac: the exact path of the pixels k_hashfilter_ac<.., DEFER> could not certify, as a small stream-ordered kernel
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_fix_ac.  The main kernel filtered every listed pixel with its APPROXIMATE bucket (hash_phase_defer); this kernel computes the
// reference's bucket of each (exact tensor in sumitup_ps_512's association, Raisr_AVX512.cpp:69-131; GetHashValue, :175-258) and,
// where it differs, redoes the pixel's filter step (DotProdPatch :134-149, accept test Raisr.cpp:1196-1200, tail re-hash rules as
// in filter_phase) and overwrites its HR value -- before k_blend reads the HR plane.  One wave per tile, entries of the tile's four
// wave regions taken 64 at a time in three phases:
//   1. exact tensors with 16 lanes per entry, four entries per round (exact_tensor16's scheme on the 13 x 13 LR window read straight
//      from the L2 / Infinity-Cache resident LR plane), handed to lane (entry index mod 64) through the LDS crossbar (ds_bpermute);
//   2. the hash with ONE LANE PER ENTRY (the ~200-instruction hash costs the same for 1 or 64 active lanes);
//   3. the entries whose bucket changed (a minority: the approximate value is on the right side of the boundary more often than
//      not), four per round with 16 lanes each: eight window samples and eight coefficients per lane, the 16-lane chains and tree.
// LDS: the 1 KB table of VRCP14 / VRSQRT14 only, so that workgroups of this kernel fit next to the main kernel's on a busy CU.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fix_ac(const T* __restrict__ lr, PassParams P, FixAc F, uint8_t* __restrict__ hash_out, float* __restrict__ hr,
                                                unsigned tile_first, unsigned tile_count)
{
    __shared__ uint2 sTab[128];
    if (threadIdx.x < 128) sTab[threadIdx.x] = P.tab14[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, l = lane & 15, lc = min(l, 10);
    const unsigned t = blockIdx.x * 4u + (unsigned)wv;
    if (t >= tile_count) return;
    const unsigned tile = tile_first + t;                      // tile id within the frame
    const unsigned region0 = (blockIdx.z * F.zs_tiles + tile) * 4u;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; hash_out += blockIdx.z * P.zs_hash;    // frame batches
    unsigned cw = *reinterpret_cast<const unsigned*>(F.counts + region0);
    cw = __builtin_amdgcn_readfirstlane(cw);
    unsigned nw[4], pre[5];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nw[k] = (cw >> (8 * k)) & 0xFFu;
        if (nw[k] == 0xFFu) nw[k] = 0;                        // that wave ran the all-exact code itself
        pre[k + 1] = pre[k] + nw[k];
    }
    const unsigned total = pre[4];
    if (total == 0) return;
    const int by = (int)(tile / (unsigned)F.tiles_x), bx = (int)(tile - (unsigned)by * (unsigned)F.tiles_x);
    const int c0 = kMargin + bx * 64, r0 = kMargin + by * 16;
    const uint16_t* list = F.entries + (size_t)region0 * kWaveCap;
    // entry e of the tile (regions concatenated): its pixel and approximate bucket
    auto fetch = [&](unsigned e, int& r, int& c, unsigned& bucket) {
        const unsigned k = (unsigned)(e >= pre[1]) + (unsigned)(e >= pre[2]) + (unsigned)(e >= pre[3]);
        const unsigned base = k == 0 ? 0u : (k == 1 ? pre[1] : (k == 2 ? pre[2] : pre[3]));
        const unsigned ent = list[k * kWaveCap + (e - base)];
        r = r0 + 4 * (int)k + (int)(ent & 3u);
        c = c0 + (int)((ent >> 2) & 63u);
        bucket = ent >> 8;
    };
    float wl[11];
#pragma unroll
    for (int i = 0; i < 11; i++) wl[i] = P.gauss_dev[lc * 12 + i];
    const unsigned pitch = (unsigned)P.lr_pitch;

    for (unsigned e0 = 0; e0 < total; e0 += 64u) {
        const unsigned nb = min(64u, total - e0);
        // ---- phase 1: exact tensors, 16 lanes per entry ----
        float A = 0.f, B = 0.f, D = 0.f;
        for (unsigned rd = 0; 4u * rd < nb; rd++) {
            const unsigned e = e0 + min(4u * rd + (unsigned)g, nb - 1u);
            int r, c; unsigned bucket;
            fetch(e, r, c, bucket);
            // column x = c - 5 + l of the window: rows r-6 .. r+6 of it, rows r-5 .. r+5 of its two neighbours
            const T* col = lr + (unsigned)(r - 6) * pitch + (unsigned)(c - 5 + lc);
            float Lc[13], Ll[11], Lr[11];
#pragma unroll
            for (int j = 0; j < 13; j++) Lc[j] = (float)col[(unsigned)j * pitch];
#pragma unroll
            for (int i = 0; i < 11; i++) {
                Ll[i] = (float)col[(unsigned)(i + 1) * pitch - 1];
                Lr[i] = (float)col[(unsigned)(i + 1) * pitch + 1];
            }
            f2 AD = {0.f, 0.f};
            float Bs = 0.f;
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const f2 gg = {Lc[i + 2] - Lc[i], Lr[i] - Ll[i]};          // GetGx: row below - row above; GetGy: right - left
                const f2 w2 = {wl[i], wl[i]};
                const f2 pq = gg * w2;
                AD = __builtin_elementwise_fma(pq, gg, AD);
                Bs = __builtin_fmaf(pq.x, gg.y, Bs);
            }
            const bool lane3 = l == 3;
            const float a = fold11(AD.x, lane3), b = fold11(Bs, lane3), d = fold11(AD.y, lane3);     // valid in lane 0 of every group
            // lane i in [4 rd, 4 rd + 4) takes the result of group i & 3
            const int src = (lane & 3) * 16;
            const float av = __shfl(a, src), bv = __shfl(b, src), dv = __shfl(d, src);
            if ((unsigned)(lane >> 2) == rd) { A = av; B = bv; D = dv; }
        }
        // ---- phase 2: the hash, one lane per entry ----
        int r = 0, c = 0;
        unsigned bucket = 0, hA = 0xFFu, hB = 0xFFu;
        bool redo = false;
        if ((unsigned)lane < nb) {
            fetch(e0 + (unsigned)lane, r, c, bucket);
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
            auto dot = [&](unsigned h) -> float {
                const float* f = P.bank + ((size_t)h * (unsigned)P.pixel_types + type) * kTapsPad + l;
                float acc = x[0] * f[0];
#pragma unroll
                for (int ch = 1; ch < 8; ch++) acc = __builtin_fmaf(x[ch], f[16 * ch], acc);
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
