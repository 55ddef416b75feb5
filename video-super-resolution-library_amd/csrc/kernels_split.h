// kernels_split.h -- split pipeline (RAISR_HIP_SPLIT=1): k_hash_ac, k_fix_sparse, k_fix_dense, k_filter_lds16 (filter bank in LDS)
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

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


