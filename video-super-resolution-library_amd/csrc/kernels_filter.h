// kernels_filter.h -- 121-tap filter stage (filter_phase), k_filter, and the fused tile kernels k_hashfilter / k_hashfilter_ac (production)
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_filter: HR = (lo < v < hi) ? v : LR with v = DotProdPatch(patch, bank[hash][type])
// (Raisr_AVX512.cpp:134-149; accept test Raisr.cpp:1196-1200).  16 lanes per pixel: lane l owns the
// reference's zmm lane l: acc = p[l]*f[l]; acc = fma(p[16c+l], f[16c+l], acc) for c=1..7;
// then sumitup_ps_512 as DPP row rotations by 8, 4, 2, 1.
// Tile = 64 columns x 16 rows; wave w owns tile rows [4w, 4w+4); each step handles 4 adjacent pixels.
// ------------------------------------------------------------------------------------------------
// Lane-major copy of the fp32 bank (built once per model): the filter stage's lane l reads taps l, 16 + l, ..., 112 + l of a
// row; stored next to each other, the four coefficients the symmetric stage needs are ONE 16-byte load per pixel step instead
// of four 4-byte loads -- the same bytes through the vector L1 with a quarter of the load instructions.
constexpr unsigned kLmRow = 128;          // floats per row of the lane-major bank
__global__ __launch_bounds__(256) void k_lane_major_bank(const float* __restrict__ bank, float* __restrict__ out, unsigned n)
{
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    const unsigned r = e >> 7, k = e & 127u, ch = k >> 4, l = k & 15u;
    out[r * 128u + (ch >> 2) * 64u + l * 4u + (ch & 3u)] = bank[e];
}

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

// Symmetric banks.  Most rows of the shipped high-resolution banks are palindromes, f[k] == f[120 - k] bit for bit (the
// trained filters are point-symmetric; 862 of 864 rows of filters_2x/filters_highres/filterbin_2_8).  For such a row the
// reference's zmm lane l (taps l, 16 + l, ..., 112 + l) and lane l' = (8 - l) & 15 need the SAME eight coefficients in opposite
// order: f[16 ch + l] = f[120 - 16 ch - l] = f[16 (7 - ch) + (8 - l)] (l <= 8), = f[16 (6 - ch) + (24 - l)] (l >= 9, whose
// eighth tap 112 + l is padding).  filter_phase<.., SYM> therefore loads only the four coefficients f[16 c + l], c = 0..3
// (the first 256 B of the row: half the bytes through the vector L1, the resource that binds this stage), runs taps
// ch = 0..3 of lane l's chain, hands the accumulator to the partner lane with two DPP moves (partner_xchg) and continues
// there with taps ch = 4..7 of chain l' = (8 - l) & 15 on the coefficients that lane already holds, in reverse order.
// Every chain still sees its eight fused multiply-adds in the reference's order; the 16 sums end up permuted by
// l -> (8 - l) & 15, which maps the summation tree of sumitup_ps_512 onto itself (it flips tree levels only), so the
// tree's result is the same bits.  For l >= 9 the chain's padding step (p = +0, f = +0: acc + (+0)) runs as the FIFTH step
// instead of the eighth so that both lane classes use the registers (c3, c2, c1, c0) for steps 4..7, and it runs as
// fma(+0, c3, acc): the lane reads its "pixel" from a block of zeros instead of masking the coefficient (one select per
// tile row instead of one per step).  Signed zeros: the product is +0 or -0 by the sign of c3, so the step leaves acc as it
// is, except that -0 + (+0) = +0.  A chain value of -0 can only come from -0 products on a -0 accumulator (an exact
// cancellation rounds to +0), so moving the step inside the chain, or skipping it, changes nothing unless the chain's
// eight products are all -0 -- then this lane ends with -0 or +0 where the reference has +0.  A zero chain adds nothing to
// a tree whose total is not zero, and a zero total fails the accept test either way (clamp_lo >= 0, checked at configure),
// so the pixel keeps LR in both cases: every stored value has the reference's bits.  (tests/test_sym_filter_model.py
// replays this lane program on the CPU against the plain 16-lane chains.)  Rows that are not palindromes (in a tap pair the stage
// mirrors: k <= 56) are listed in P.asym; the steps that hold their pixels run a second time with the true coefficients (below).
__device__ __forceinline__ float partner_xchg(float v)      // lane p of every row of 16 receives lane (8 - p) & 15
{
    // (every lane of a row has a source lane: `old` is never used, so it is the source itself and no register is zeroed for it)
    const int x = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, true);                      // row_mirror: t[p] = v[15 - p]
    t = __builtin_amdgcn_update_dpp(t, t, 0x129, 0xf, 0xf, true);                          // row_ror:9 : lane p reads lane (p - 9) & 15
    return __int_as_float(t);
}

// filter_phase: the work of one 64 x 16 tile once its LR window is in LDS -- sP points at window position
// (row r0-5, column c0-5), row stride LW -- and the tile's hashes are in sH / sH2 (0xFF = not filtered / no re-hash).
//
// PC ("pair columns", the production kernel): group g of step s = 2 p + j filters column 8 p + 2 g + j instead of 4 s + g, so the two
// steps of a pair need ADJACENT window values per tap: one ds_read_b64 -- 2.8 LDS cycles per wave-instruction -- instead of one
// ds_read2_b32 (two accesses 16 B apart: 4.7; scripts/lds_b64_probe.hip, profiles/r05_lds_b64_probe.log).  gfx950 serves an 8-byte
// read at a 4-mod-8 address correctly but lane by lane (64 cycles), so every read must be aligned: the window's row stride is even
// (LW = 76), which makes the alignment of tap (i, j) of an even column a function of j alone, and the taps that would be misaligned in
// the window (even j: sP sits at an odd dword) read a second copy sQ of the filter window whose origin is aligned (sQ[e] == sP[e],
// 26 x 76 floats; the caller builds it in LDS that is dead by then).  The type column (c - 5) & 1 = (j + 1) & 1 (c0 is even) no longer
// depends on the lane: it travels in the coefficient loads' scalar offset.  Same operations on the same operands: the pixel -> lane
// group assignment is free.
template <int LW, int RPW = 4, bool SYM = false, bool PC = false>
__device__ __forceinline__ void filter_phase(const PassParams& P, const float* sL, const uint8_t* sH, const uint8_t* sH2,
                                             int c0, int r0, float* __restrict__ hr, unsigned tid = threadIdx.x, const float* zpad = nullptr,
                                             const float* sQ = nullptr)
{
    constexpr int TW = 64;
    static_assert(!PC || (LW % 2) == 0, "pair columns: 8-byte window reads need an even row stride");
    const int lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l = lane & 15;
    int off[8];
    const int copy_disp = PC ? (int)(sQ - sL) : 0;               // floats from a window position to the same position of the second copy
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
        int k = 16 * ch + l;
        if (SYM && ch >= 4) {                                   // steps 4..7 run the partner's chain l2 on this lane
            const int l2 = (8 - l) & 15;
            k = l <= 8 ? 16 * ch + l2 : (ch == 4 ? kTaps : 16 * (ch - 1) + l2);   // l >= 9: padding step first, then taps ch = 4, 5, 6
        }
        off[ch] = (k < kTaps) ? (k / 11) * LW + (k % 11) : 0;   // padding taps: coefficient is +0, any finite pixel will do
        if (PC && (k >= kTaps || ((k % 11) & 1) == 0)) off[ch] += copy_disp;      // even patch column (and the padding taps): the aligned copy
        asm volatile("" : "+v"(off[ch]));                      // one register per tap: left alone, the compiler keeps row and column part apart (16 VGPRs)
    }
    // SYM: step 4 is tap 64 + l2 times c3 for l <= 8 and the padding step for l >= 9, whose window address points at zpad (64 floats
    // of +0 owned by this wave): p = +0 times c3 -- see the note on signed zeros above.

    // 32-bit buffer addressing of the filter bank (one descriptor per wave, built from uniform values).  The stage reads the
    // lane-major copy of the bank (k_lane_major_bank): the lane's coefficients of taps ch = 0..3 are one 16-byte load, ch = 4..7 the
    // one 256 B further on.
    const __amdgpu_buffer_rsrc_t bank_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.bank_lm), 0, P.bank_bytes, 0x00020000);
    const int tcol = (!PC && P.pixel_types == 4) ? ((g + 1) & 1) : 0;        // (c-5)&1 with c = c0 + 4s + g, c0 even
    const unsigned lane_off = (unsigned)(tcol * kTapsPad + 4 * l) * 4u;      // byte offset of (type column part, zmm lane)
    const unsigned tsoff[2] = {(PC && P.pixel_types == 4) ? (unsigned)(kTapsPad * 4) : 0u, 0u};     // PC: step parity j -> type column part
#define RAISR_COL(s) (PC ? 8 * ((s) >> 1) + 2 * g + ((s) & 1) : 4 * (s) + g)
#define RAISR_G0 (PC ? 2 * g : g)
    const unsigned bank_stride = (unsigned)(P.pixel_types * kTapsPad * 4);   // bytes per hash bucket (<= 2048)
    // Symmetric stage, true coefficients of the taps behind the hand-over (second runs of the steps that hold pixels of non-palindromic rows): lane l2 = (8 - l) & 15's run of taps
    // 64 + l2 .. 112 + l2 in the lane-major row, 256 + 16 l2 bytes into it.  Lanes >= 9 multiply the padding step first and then taps
    // 64 + l2, 80 + l2, 96 + l2: they load from 4 bytes EARLIER, so that the same registers serve both lane classes -- the float in front
    // (tap 111 + l2: tap 120 or padding zeros) meets the +0 of the zero block (any finite coefficient will do there, see above; a 16-byte
    // buffer load needs 4-byte alignment only, and none of these straddles a 128-byte line).
    const unsigned poff = 256u + 16u * (unsigned)((8 - l) & 15) - 16u * (unsigned)l - (l >= 9 ? 4u : 0u);

#pragma unroll 1                                                 // (unrolled 2x / 4x: no difference, r04_call16)
    for (int row = 0; row < RPW; row++) {
        const int prow = RPW * w + row;
        const int r = r0 + prow;
        const unsigned trow_off = (P.pixel_types == 4) ? (unsigned)(((r - 5) & 1) * 2 * kTapsPad * 4) : 0u;
        const unsigned row_lane_off = trow_off + lane_off;
        // LDS byte addresses of this lane's 8 taps (and the centre pixel) for step 0; step s adds the immediate 16*s
        const char* tap[8];
#pragma unroll
        for (int ch = 0; ch < 8; ch++) tap[ch] = reinterpret_cast<const char*>(sL + prow * LW + RAISR_G0 + off[ch]);
        if (SYM) tap[4] = l >= 9 ? reinterpret_cast<const char*>(zpad) : tap[4];
        const char* ctr = reinterpret_cast<const char*>(sL + prow * LW + RAISR_G0 + 5 * LW + 5);
#ifdef RAISR_PROBE_LDS_F                                         /* development builds: timing probe from kernels_probes.h */
#define RAISR_LDS_F(p, s) RAISR_PROBE_LDS_F(p, s)
#else
#define RAISR_LDS_F(p, s) (*reinterpret_cast<const float*>((p) + (PC ? 32 * ((s) >> 1) + 4 * ((s) & 1) : 16 * (s))))
#endif
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RAISR_BANK_F4(voff, s) __builtin_amdgcn_raw_buffer_load_b128(bank_rsrc, (voff), PC ? tsoff[(s) & 1] : 0u, 0)
        // The plain 16-lane chains of one step with all eight coefficients (tail re-hash and, in the symmetric variant, the pixels of
        // non-palindromic rows).  The symmetric variant has no registers for the plain tap offsets: it recomputes them here, behind
        // an opaque lane index so that they are not hoisted into the main loop's live range.
        auto plain_step = [&](int s, unsigned hb) -> float {
            const unsigned voff = __umul24(hb, bank_stride) + row_lane_off;
            const u32x4 fa = RAISR_BANK_F4(voff, s), fb = RAISR_BANK_F4(voff + 256u, s);
            float acc;
            if (!SYM) {
                acc = RAISR_LDS_F(tap[0], s) * __uint_as_float(fa[0]);
#pragma unroll
                for (int ch = 1; ch < 8; ch++) acc = __builtin_fmaf(RAISR_LDS_F(tap[ch], s), __uint_as_float(ch < 4 ? fa[ch & 3] : fb[ch & 3]), acc);
            } else {
                int lq = l;
                asm volatile("" : "+v"(lq));
                const float* base = sL + prow * LW + RAISR_COL(s);
                acc = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 8; ch++) {
                    const int k = 16 * ch + lq;
                    const float x = base[(k < kTaps) ? (k / 11) * LW + (k % 11) : 0];
                    const float f = __uint_as_float(ch < 4 ? fa[ch & 3] : fb[ch & 3]);
                    acc = ch == 0 ? x * f : __builtin_fmaf(x, f, acc);
                }
            }
            return tree16(acc);
        };
        const bool anyB = sH2[prow * TW + lane] != 0xFFu;      // does this tile row contain re-hashed (tail) columns?
        // symmetric stage: is this lane's pixel (column c0 + lane) in a non-palindromic bank row?  The bitmap word is requested here and
        // used after the row's steps -- asked for there, its global round trip ended every row with an exposed s_waitcnt vmcnt(0)
        // (round 5, R5.10: C2 +1.3 %)
        unsigned asym_word = 0u, asym_key = 0u;
        if (SYM && P.asym) {
            const unsigned hrow = sH[prow * TW + lane];
            asym_key = __umul24(hrow, (unsigned)P.pixel_types) + ((P.pixel_types == 4) ? (unsigned)(((r - 5) & 1) * 2 + ((lane + 1) & 1)) : 0u);
            if (hrow != 0xFFu) asym_word = P.asym[asym_key >> 5];
        }
        // The row's 16 steps (4 adjacent pixels each) share ONE summation tree.  sumitup_ps_512 halves the number of distinct
        // values per step at every level (16 lanes -> r8[0..7] -> r4[0..3] -> r2[0..1] -> v), so after each level two steps are
        // merged into one register (v_cndmask on a lane-index bit) and the next level runs once for both: 16 + 8 + 4 + 2 adds
        // and 8 + 4 + 2 + 1 merges per row, against 16 x 4 adds when every step folds on its own.  Same additions, same
        // operands (level 2 uses row_shl:4 / row_shr:4 so that lanes pair inside their half of the row; a + b == b + a bit for
        // bit).  Lane l ends up with the result of step sl = bitrev4(l): merge level 1 puts step bit 0 on lane bit 3, ..., level 4
        // step bit 3 on lane bit 0.
        float A16[16];
        // One step: the lane's chain.  No branch for a bucket byte of 0xFF (pixel not filtered): its offset lies past the bank, the
        // bounds-checked buffer loads return +0, v = 0 fails the accept test (clamp_lo >= 0, checked at configure) and the pixel keeps LR.
        auto chain = [&](const float (&x)[8], const float (&q)[8]) -> float {
            float acc = x[0] * q[0];
            if (!SYM) {
#pragma unroll
                for (int ch = 1; ch < 8; ch++) acc = __builtin_fmaf(x[ch], q[ch], acc);
            } else {                                            // q[0..3] only: the partner chain runs on the same four, backwards
                acc = __builtin_fmaf(x[1], q[1], acc);
                acc = __builtin_fmaf(x[2], q[2], acc);
                acc = __builtin_fmaf(x[3], q[3], acc);
                acc = partner_xchg(acc);
                acc = __builtin_fmaf(x[4], q[3], acc);
                acc = __builtin_fmaf(x[5], q[2], acc);
                acc = __builtin_fmaf(x[6], q[1], acc);
                acc = __builtin_fmaf(x[7], q[0], acc);
            }
            return acc + row_ror<0x128>(acc);                   // r8[i] = a[i] + a[i+8]: lanes i and i ^ 8 hold the same value
        };
        // symmetric lane program on both runs of four (second runs): q[4..7] = the partner's run (lanes >= 9: from 4 bytes earlier)
        auto chain_full = [&](const float (&x)[8], const float (&q)[8]) -> float {
            float acc = x[0] * q[0];
            acc = __builtin_fmaf(x[1], q[1], acc);
            acc = __builtin_fmaf(x[2], q[2], acc);
            acc = __builtin_fmaf(x[3], q[3], acc);
            acc = partner_xchg(acc);
#pragma unroll
            for (int ch = 4; ch < 8; ch++) acc = __builtin_fmaf(x[ch], q[ch], acc);
            return acc + row_ror<0x128>(acc);
        };
        auto load_q_full = [&](unsigned hb, float (&q)[8], int s) {
            const unsigned voff = __umul24(hb, bank_stride) + row_lane_off;
            const u32x4 fa = RAISR_BANK_F4(voff, s), fb = RAISR_BANK_F4(voff + poff, s);
#pragma unroll
            for (int ch = 0; ch < 4; ch++) { q[ch] = __uint_as_float(fa[ch]); q[4 + ch] = __uint_as_float(fb[ch]); }
        };
        auto load_q = [&](unsigned hb, float (&q)[8], int s) {    // v_mad_u32_u24 for the offset (the 32x32 form is a slow 64-bit mad)
            const unsigned voff = __umul24(hb, bank_stride) + row_lane_off;
            const u32x4 fa = RAISR_BANK_F4(voff, s);
#pragma unroll
            for (int ch = 0; ch < 4; ch++) q[ch] = __uint_as_float(fa[ch]);
            if (!SYM) {
                const u32x4 fb = RAISR_BANK_F4(voff + 256u, s);
#pragma unroll
                for (int ch = 0; ch < 4; ch++) q[4 + ch] = __uint_as_float(fb[ch]);
            }
        };
#ifdef RAISR_PROBE_FILTER_STEPS                                  /* development builds: timing probe from kernels_probes.h (output wrong) */
        RAISR_PROBE_FILTER_STEPS
#else
        {
            // The steps, software-pipelined by hand in pairs (a ds_read2_b32 fetches one tap of two steps): the coefficient loads and
            // window reads of pair p + 1 are issued before the arithmetic of pair p, the bucket bytes one pair earlier still, with a
            // scheduling fence per pair.  Against the compiler's own schedule (loads six steps ahead, window reads just in time):
            // symmetric stage +0.8 % with one or two pairs of look-ahead, +0 % with three (r04_call19); the eight-load stage +1 %
            // with 4-byte loads and +3.6-4.7 % with the two 16-byte loads (C1, C5; r04_call23), which on the compiler's schedule
            // were 4 % SLOWER than eight 4-byte loads (r04_call18).  Same operations on the same operands.
            constexpr int AHEAD = 1;
            float Q[16][8];
            float X[16][8];
            unsigned Hh[16];
            // PC: the bucket bytes of a pair are adjacent -- one 2-byte read, unpacked where the coefficient loads are issued (pinned there:
            // taken apart right behind the read, the unpacking would wait for it and expose the LDS latency in every pair).  Two 1-byte
            // reads instead: -0.3 % on C2; two pairs of look-ahead (AHEAD = 2, 102 VGPRs): +-0 (profiles/r05_paircol_variants_ab.log)
            auto issue_h = [&](int s) {
                if constexpr (PC) { if (!(s & 1)) Hh[s] = *reinterpret_cast<const uint16_t*>(sH + prow * TW + RAISR_COL(s)); }
                else Hh[s] = sH[prow * TW + RAISR_COL(s)];
            };
            auto bucket_of = [&](int s) -> unsigned {
                if constexpr (PC) {
                    if (!(s & 1)) { asm volatile("" : "+v"(Hh[s])); return Hh[s] & 0xFFu; }
                    return Hh[s - 1] >> 8;
                } else return Hh[s];
            };
            // the window values of both steps of pair p.  PC: ONE aligned 8-byte LDS read per tap (the cast states the alignment the two-copy
            // layout guarantees; left to itself the compiler assumes 4 and emits ds_read2_b32 offset1:1)
            typedef float f32x2a8 __attribute__((ext_vector_type(2), aligned(8)));
            auto issue_xx = [&](int p) {
#pragma unroll
                for (int ch = 0; ch < 8; ch++) {
                    if constexpr (PC) {
                        const f32x2a8 v = *reinterpret_cast<const f32x2a8*>(tap[ch] + 32 * p);
                        X[2 * p][ch] = v.x; X[2 * p + 1][ch] = v.y;
                    } else { X[2 * p][ch] = RAISR_LDS_F(tap[ch], 2 * p); X[2 * p + 1][ch] = RAISR_LDS_F(tap[ch], 2 * p + 1); }
                }
            };
#pragma unroll
            for (int s = 0; s < 2 * AHEAD + 2; s++) issue_h(s);
#pragma unroll
            for (int s = 0; s < 2 * AHEAD; s++) load_q(bucket_of(s), Q[s], s);
            issue_xx(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 8; p++) {
                if (p + AHEAD < 8) { load_q(bucket_of(2 * (p + AHEAD)), Q[2 * (p + AHEAD)], 0); load_q(bucket_of(2 * (p + AHEAD) + 1), Q[2 * (p + AHEAD) + 1], 1); }
                if (p + AHEAD + 1 < 8) { issue_h(2 * (p + AHEAD + 1)); issue_h(2 * (p + AHEAD + 1) + 1); }
                if (p + 1 < 8) issue_xx(p + 1);
                __builtin_amdgcn_sched_barrier(0);
                A16[2 * p] = chain(X[2 * p], Q[2 * p]);
                A16[2 * p + 1] = chain(X[2 * p + 1], Q[2 * p + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        // Symmetric stage, pixels of non-palindromic bank rows (round 6; before: the pixel was redone after the accept test with a plain
        // 16-lane step -- eight recomputed window addresses, its own summation tree, its own accept test).  A step that holds such a pixel
        // runs AGAIN here, before the row's shared summation tree, with the same lane program on the TRUE coefficients: what a lane
        // multiplies after the hand-over -- taps 64 + l2, 80 + l2, 96 + l2 (, 112 + l2) of the partner chain l2 = (8 - l) & 15 -- is lane
        // l2's second run of four in the lane-major bank row (256 B + 16 l2), so ONE more 16-byte load replaces the mirrored registers
        // (lanes >= 9 load from 4 bytes earlier: `poff` above).  The window reads use the loop's own tap addresses with the step's
        // immediate offset: the steps are unrolled behind wave-uniform branches on the ballot of the row's 64 pixels (PC: a step's pixels
        // are columns 8 p + j + {0, 2, 4, 6}).  The step's other pixels are recomputed to the same bits (palindromic row: the loaded run
        // equals the mirrored registers; unfiltered pixel: both loads return +0), so A16[s] is simply replaced and tree, accept test and
        // store stay shared.  (tests/test_sym_filter_model.py replays this lane program on rows with arbitrary taps.)
        if (SYM && P.asym) {
            const unsigned long long am = __ballot((asym_word >> (asym_key & 31u)) & 1u);      // (0 for a pixel that is not filtered: its word was not loaded)
            if (am) {
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    const unsigned long long sm = PC ? (0x55ull << (8 * (s >> 1) + (s & 1))) : (0xFull << (4 * s));
                    if ((am & sm) == 0) continue;
                    float q[8], x[8];
                    load_q_full(sH[prow * TW + RAISR_COL(s)], q, s);
#pragma unroll
                    for (int ch = 0; ch < 8; ch++) x[ch] = RAISR_LDS_F(tap[ch], s);
                    A16[s] = chain_full(x, q);
                }
            }
        }
#define RAISR_MERGE(dst, src, mask) asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask))
        float B8[8], C4[4], D2[2];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float t = A16[2 * k];
            RAISR_MERGE(t, A16[2 * k + 1], 0xff00ff00ff00ff00ull);            // lanes 8..15 of every row: the odd step
            // r4[i] = r8[i] + r8[i+4] inside each half; even k keeps lanes {0-3, 8-11} (row_shl:4), odd k lanes {4-7, 12-15} (row_shr:4)
            B8[k] = (k & 1) ? t + row_ror<0x114>(t) : t + row_ror<0x104>(t);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float t = B8[2 * m];
            RAISR_MERGE(t, B8[2 * m + 1], 0xf0f0f0f0f0f0f0f0ull);
            C4[m] = t + quad_perm<0x4e>(t);                    // [2,3,0,1]: r2 = r4[i] + r4[i+2]
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            float t = C4[2 * n];
            RAISR_MERGE(t, C4[2 * n + 1], 0xccccccccccccccccull);
            D2[n] = t + quad_perm<0xb1>(t);                    // [1,0,3,2]: r2[0] + r2[1]
        }
        float v = D2[0];
        RAISR_MERGE(v, D2[1], 0xaaaaaaaaaaaaaaaaull);
#undef RAISR_MERGE
        const int sl = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);     // the step whose pixel this lane keeps
        const int csl = RAISR_COL(sl) - RAISR_G0;                                              // the pixel's column in the tile, less the group's base
        float keep = *reinterpret_cast<const float*>(ctr + 4 * csl);
        if (v > P.lo && v < P.hi) keep = v;
        if (__any(anyB)) {                                      // tail columns only: AVX2 re-hash (keep-first-if-rejected;
#pragma unroll 1                                                 //  Randomness blends the last candidate instead)
            for (int s = 0; s < 16; s++) {
                const unsigned hB = sH2[prow * TW + RAISR_COL(s)];
                if (hB == 0xFFu) continue;
                const float v = plain_step(s, hB);
                if (s == sl) {
                    if (v > P.lo && v < P.hi) keep = v;
                    else if (P.randomness) keep = RAISR_LDS_F(ctr, s);
                }
            }
        }
#undef RAISR_LDS_F
#undef RAISR_BANK_F4
        const int c = c0 + RAISR_G0 + csl;
        if (r < P.H - kMargin && c < P.c_final) hr[(size_t)r * P.hr_pitch + c] = keep;
    }
#undef RAISR_COL
#undef RAISR_G0
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
// PART (development builds, RAISR_HIP_AC_PART): 0 = the production kernel, 1 = hash stage only, 2 = filter stage only.
// Four workgroups per CU (16 waves): <= 128 VGPRs and <= 40 960 B of LDS each (4 x 40 960 = the CU's 160 KB; round 6: 100 / 108 VGPRs, 40 912 B).  Round 2 ran three (144 VGPRs,
// 41 408 B): the fourth costs nothing but the two measures below -- the exact path's approximation table shares sV's space, and the
// filter stage's tap offsets are kept as ONE register each -- and buys 12 % (1080p -> 4K: 192 -> 170 us isolated).
// one 64 x 16 tile (tile column bx, tile row by) of k_hashfilter_ac: LR window -> gradient tile -> certified hash stage -> filter stage
// PC (pair columns, filter_phase): sQ is the place of the filter stage's second window copy -- inside sV's space, past the exact path's
// table and tensors, dead once every wave has finished its H pass; copy_window fills it behind the hash stage's barrier.
template <int LW>
__device__ __forceinline__ void copy_window(const float* sL, float* sQ, unsigned tid)
{
#pragma unroll
    for (unsigned i = tid; i < 26u * LW; i += 256u) sQ[i] = sL[LW + 1 + i];      // sQ[e] = sP[e], sP = sL + LW + 1: window position (r0 - 5, c0 - 5)
}

template <typename T, int PART, int LW, int LH, int GW_, int GH, typename GT, int RPW = 4, bool SYM = false, bool DEFER = false, bool PC = false>
__device__ __forceinline__ void hashfilter_ac_tile(const T* __restrict__ lr, const PassParams& P, const GaussW& gw, const SepW& S,
                                                   uint8_t* __restrict__ hash_out, float* __restrict__ hr, int bx, int by,
                                                   float* sL, GT* sG, typename FVec<RPW>::type* sV, uint2* sTab, uint8_t* sH, uint8_t* sH2, uint16_t* sList, unsigned* sCnt, unsigned tid = threadIdx.x,
                                                   const FixAc& F = FixAc{}, unsigned tile_id = 0, float* sQ = nullptr)
{
    constexpr int TW = 64, TH = 4 * RPW;
    static_assert(LH == TH + 12 && GH == TH + 10, "window and gradient tile follow the tile height");
    const int lane = tid & 63, w = tid >> 6;
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;

    RAISR_PHASE_DECL;
    if (!DEFER && tid < 4) sCnt[tid] = 0;
    stage_tile<LH, 76, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL, tid);
    RAISR_BARRIER(tid);
    RAISR_PHASE(0);
    unsigned any_grad = 0u;                                    // OR of the bit patterns of this thread's gradients (x - x is +0: all zero bits <=> flat)
    {   // gradient tile: G(ty,tx) <-> image (r0-5+ty, c0-5+tx) <-> L tile (ty+1, tx+1)
        auto grad = [&](int ty, int tx) {
            const float gxv = sL[(ty + 2) * LW + tx + 1] - sL[ty * LW + tx + 1];
            const float gyv = sL[(ty + 1) * LW + tx + 2] - sL[(ty + 1) * LW + tx];
            any_grad |= __float_as_uint(gxv) | __float_as_uint(gyv);      // one v_or3_b32
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
            const unsigned idx = tid + 256u * it;
            const int ty = (int)(idx / (GW_ - 64)), tx = 64 + (int)(idx - (unsigned)ty * (GW_ - 64));
            if (idx < NR) grad(ty, tx);
        }
    }
    if (!DEFER && __any(any_grad != 0u) && lane == 0) sCnt[3] = 1u;  // one LDS store per wave that saw a gradient
    RAISR_BARRIER(tid);
    RAISR_PHASE(1);
    // FLAT TILE: not one non-zero gradient in the whole 26 x 74 gradient tile -- letterbox bars, flat graphics, fades to black.  Every
    // window of the tile is flat, so the reference's tensor is exactly (0, 0, 0) for every pixel (as in the per-pixel `zero` case of the
    // hash stage) and the buckets are the zero tensor's: the separable passes, the hash and the worklist are skipped (a quarter of the
    // tile's work); the filter stage runs as always.  (Not in self-check mode, which wants every pixel through both paths.)
    const bool flat_tile = !DEFER && sCnt[3] == 0u && !P.cert_check;
#ifdef RAISR_HIP_DEV
    if (PART == 2) {     // profiling aid: filter stage only; P.cert_check doubles as the bucket pattern (0 = every row of the bank, 1 = one row, 2 = sixteen rows)
        for (int i = tid; i < TH * TW; i += 256) { sH[i] = (uint8_t)(P.cert_check == 1 ? 0 : (P.cert_check == 2 ? (i * 7) % 16 : (i * 7) % 216)); sH2[i] = 0xFFu; }
        if constexpr (PC) copy_window<LW>(sL, sQ, tid);
        RAISR_BARRIER(tid);
    } else
#endif
    if constexpr (DEFER) {
        static_assert(RPW == 4, "the deferred exact path is built for the 64 x 16 tile");
        hash_phase_defer<LW, GT>(P, gw, S, sG, sV, sH, sH2, F, 4u * tile_id + (unsigned)w, c0, r0, tid);
        __builtin_amdgcn_wave_barrier();                    // rows [4w, 4w+4) of sH / sH2 are written and read by wave w only; LDS is in order within a wave
    } else if (flat_tile) {
        const int c = c0 + lane;
        const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
#pragma unroll
        for (int j = 0; j < RPW; j++) {                      // rows [RPW w, RPW w + RPW) of sH / sH2 are written and read by wave w only
            const int prow = RPW * w + j;
            const bool zone = r0 + prow < P.H - kMargin && c < P.c_final && (inA || inB);
            sH[prow * TW + lane] = zone ? (uint8_t)P.zero_bucket[inA ? 0 : 1] : (uint8_t)0xFFu;
            sH2[prow * TW + lane] = (zone && inA && inB) ? (uint8_t)P.zero_bucket[1] : (uint8_t)0xFFu;
        }
        if constexpr (PC) { copy_window<LW>(sL, sQ, tid); RAISR_BARRIER(tid); }      // (flat_tile is the same in every thread)
        else __builtin_amdgcn_wave_barrier();
    } else
    hash_phase_ac<LW, GT, RPW, PC>(P, gw, S, sL, sG, sV, sTab, sH, sH2, sList, sCnt, c0, r0, tid, sQ);
    RAISR_PHASE_RESET;                                     // (the hash stage keeps its own marks 2..5)
    if (P.write_hash) {
        const int c = c0 + lane;
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = r0 + RPW * w + j;
            if (r < P.H - kMargin && c < P.c_final) hash_out[(unsigned)r * (unsigned)P.hash_pitch + (unsigned)c] = sH[(RPW * w + j) * TW + lane];
        }
    }
    if (P.cert_stats && tid == 0) {
        const int zr = min(TH, P.H - kMargin - r0), zc = min(TW, P.c_final - c0);
        if (!DEFER && sCnt[1]) atomicAdd(&P.cert_stats[0], sCnt[1]);
        if (!DEFER && sCnt[2]) atomicAdd(&P.cert_stats[1], sCnt[2]);
        atomicAdd(&P.cert_stats[2], (unsigned)(max(zr, 0) * max(zc, 0)));
        // per-tile view of the worklist (meaningful without the self-check, which lists every pixel): tiles with a non-empty list, tiles
        // whose list overflowed (they pay the approximate AND the all-exact stage), tiles seen, flat tiles (hash stage skipped)
        if (!DEFER && !flat_tile && sCnt[0]) atomicAdd(&P.cert_stats[3], 1u);
        if (!DEFER && !flat_tile && sCnt[0] > kListMax) atomicAdd(&P.cert_stats[4], 1u);
        atomicAdd(&P.cert_stats[5], 1u);
        if (flat_tile) atomicAdd(&P.cert_stats[6], 1u);
    }
    // symmetric stage: 64 floats of +0 per wave in the gradient tile's space (every wave is past its last read of sG: the hash
    // stage's last use of it lies before a workgroup barrier); written and read by the same wave, LDS operations of a wave run in order.
    // Deferred variant: the gradient tile stays live while any wave may still take its all-exact fallback, so the block lives in the
    // wave's own rows of sV (row group w of channel 0: read by wave w's H pass only, which is over).
    float* zpad = DEFER ? reinterpret_cast<float*>(sV + w * GW_) : reinterpret_cast<float*>(sG) + 64 * w;
    if (SYM && PART != 1) zpad[lane] = 0.0f;
    if (PART != 1) filter_phase<LW, RPW, SYM, PC>(P, sL + LW + 1, sH, sH2, c0, r0, hr, tid, zpad, sQ);
    else if (sH[tid & (TH * TW - 1)] == 0xFEu) hr[0] = 0.f;       // keep the hash stage alive
    RAISR_PHASE(6);
}

template <typename T, int PART = 0, int RPW = 4, bool SYM = false, bool DEFER = false>
#ifdef RAISR_EXP_OCC5
#define RAISR_AC_WGS 5
#else
#define RAISR_AC_WGS 4
#endif
__global__ __launch_bounds__(256, RPW == 4 ? RAISR_AC_WGS : 6) void k_hashfilter_ac(const T* __restrict__ lr, PassParams P, GaussW gw, SepW S,
                                                                         uint8_t* __restrict__ hash_out, float* __restrict__ hr, FixAc F = FixAc{})
{
    constexpr int TW = 64, TH = 4 * RPW;
    // pair columns (filter_phase): the production variants; the deferred comparison pipeline and the 64 x 8 experiment keep the round-4 order
    constexpr bool PC = RPW == 4 && !DEFER;
    constexpr int LW = PC ? 78 : 77, LH = TH + 12, GW_ = 74, GH = TH + 10;
    using GT = typename GradOf<T>::type;
    using VT = typename FVec<RPW>::type;
    // One block of LDS, carved by hand: the second window copy's bank alignment against the window (below) needs known offsets.
    constexpr unsigned oL = 0, oG = oL + LH * LW * 4, oV = oG + GH * GW_ * sizeof(GT), oH = oV + 3 * 4 * GW_ * sizeof(VT), oH2 = oH + TH * TW,
                       oList = oH2 + TH * TW, oCnt = oList + (DEFER ? 1 : kListMax) * 2 + (DEFER ? 2 : 0), oEnd = oCnt + (DEFER ? 1 : 4) * 4;
    static_assert(oG % 8 == 0 && oV % 16 == 0 && oCnt % 4 == 0, "LDS carving: alignment");
    __shared__ __attribute__((aligned(256))) unsigned char smem[oEnd];
    float* sL = reinterpret_cast<float*>(smem + oL);
    GT* sG = reinterpret_cast<GT*>(smem + oG);
    VT* sV = reinterpret_cast<VT*>(smem + oV);
    uint2* sTab = reinterpret_cast<uint2*>(sV);   // the exact path's table takes sV's place once the H pass is done (hash_phase_ac)
    uint8_t* sH = smem + oH;
    uint8_t* sH2 = smem + oH2;
    uint16_t* sList = reinterpret_cast<uint16_t*>(smem + oList);      // worklist entries   (deferred variant: the list lives in global memory, FixAc)
    unsigned* sCnt = reinterpret_cast<unsigned*>(smem + oCnt);        // worklist length; uncertain pixels; certified-but-wrong (check mode); tile has a non-zero gradient
    // Second window copy of the pair-column filter stage: in sV's space behind the exact path's table (1 KB) and tensors (kListMax x 16 B),
    // 26 x LW floats.  Its dword offset from the window, mod 64, decides the bank conflicts of the stage's 8-byte reads (two groups of 32
    // lanes, 64 banks): with LW = 78 the offsets 0, 2, 40, 42 are conflict-free for every tap chunk of both stage variants
    // (tests/test_filter_window_banks.py replays the bank arithmetic); 5216 B into sV gives 40 (3680 with the 160-entry list of rounds 3-5).
    constexpr unsigned oQ = oV + 5216;
    static_assert(!PC || (oQ >= oV + 1024 + kListMax * 16 && oQ + 26 * LW * 4 <= oH && oQ % 8 == 0), "second window copy: inside sV, past table and tensors");
    static_assert(!PC || (((oQ - oL) / 4) % 64 == 40), "second window copy: conflict-free bank offset");
    float* sQ = PC ? reinterpret_cast<float*>(smem + oQ) : nullptr;

    int bx, by;
    xcd_tile(bx, by);
    by += P.tile_y0;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; hash_out += blockIdx.z * P.zs_hash;    // frame batches
    const unsigned tile_id = blockIdx.z * F.zs_tiles + (unsigned)by * (unsigned)F.tiles_x + (unsigned)bx;
    hashfilter_ac_tile<T, PART, LW, LH, GW_, GH, GT, RPW, SYM, DEFER, PC>(lr, P, gw, S, hash_out, hr, bx, by, sL, sG, sV, sTab, sH, sH2, sList, sCnt, threadIdx.x, F, tile_id, sQ);
}



