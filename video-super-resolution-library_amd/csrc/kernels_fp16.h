// kernels_fp16.h -- gfx950 kernels reproducing the reference's AVX512-FP16 pipeline
// (ASMType AVX512_FP16; Library/Raisr_AVX512FP16.cpp) in IEEE binary16 arithmetic.
// Included by device_abi.hip inside its anonymous namespace (after kernels_hash_certify.h).
//
// Every arithmetic statement below is ONE binary16 operation with round-to-nearest-even and
// subnormals preserved (v_*_f16 / v_pk_*_f16 under the default gfx9 float mode), in the order of the
// cited reference lines.  Two things are NOT left to the compiler:
//   * division: hipcc's native f16 division (v_rcp_f32 + v_div_fixup_f16) is not correctly rounded,
//     so h_div() forces an IEEE fp32 division of the widened operands and rounds once to binary16
//     (innocuous double rounding: 24 >= 2*11+2) -- what VDIVPH produces.
//   * VRCPPH/VRSQRTPH: exponent-separable 1024-entry tables captured from Intel hardware
//     (x86_fp16_tables.h), staged in LDS.
#pragma once

typedef _Float16 hf;
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));

struct GaussW16 {
    uint32_t wT[11][12];         // wT[k][i] = (w,w) as packed binary16: un-normalised Gaussian, Raisr_globals.h:267-278
};

struct Pass16 {
    const uint32_t* bank16;      // [hash][type][4 chunks][16 lanes] half2 = (f[32c+l], f[32c+l+16]), taps >= 121 are +0
    int bank16_bytes;            // size of the binary16 bank (buffer-descriptor range)
    const uint16_t* tab16;       // rcpph T[1024], rsqrtph T0[1024], T1[1024], then the composite VRCPPH(VRSQRTPH(.)) table C[2][1024] (sqrt_ph)
    uint16_t qangle, qs0, qs1, qc0, qc1;   // binary16 bit patterns
    float nf;                    // NF_8 (fp32), Raisr_globals.h:208
    int c_avx;                   // first column of the blend stage's scalar (fp32) tail
    // Folded thresholds (fold16_thresholds, device_abi.hip): two of the hash's three binary16 divisions only feed comparisons
    // with constants, and x -> fl16(x / c) is monotone, so the comparisons are made on the dividends instead:
    //   [qs <= fl16(L1 / 100)]  ==  [L1 >= ls]         ls = the smallest binary16 L1 that passes (NaN bits: none does)
    //   [qc <= fl16(n / d)]     ==  [n >= cm d] or [n > cm d]   for d > 0, cm = the lower rounding boundary of qc (the midpoint
    //                               below it; exact in fp32, and so is cm d: 12 + 11 significant bits), ct = 1 when the
    //                               midpoint itself rounds to qc (ties to even)
    // fold = 0 (thresholds that are not positive finite numbers): the divisions stay.
    uint16_t ls0, ls1;
    float cm0, cm1;
    int ct0, ct1;
    int fold;
};

__device__ __forceinline__ hf h_bits(uint16_t u) { return __builtin_bit_cast(hf, u); }
__device__ __forceinline__ uint16_t h_u(hf h) { return __builtin_bit_cast(uint16_t, h); }

__device__ __forceinline__ hf h_div(hf a, hf b)
{
    float fa = (float)a, fb = (float)b;
    asm volatile("" : "+v"(fa), "+v"(fb));      // keep hipcc from folding this back into its fast f16 division
    float q = fa / fb;
    asm volatile("" : "+v"(q));
    return (hf)q;
}

// (float)x * normal, stored back to binary16 (Raisr_AVX512FP16.cpp:197-221)
__device__ __forceinline__ hf h_scale_f32(hf x, float nf)
{
    float f = (float)x;
    asm volatile("" : "+v"(f));
    float p = f * nf;
    asm volatile("" : "+v"(p));
    return (hf)p;
}

// VRSQRTPH / VRCPPH models (tables in LDS: T at [0,1024), T0 at [1024,2048), T1 at [2048,3072))
__device__ __forceinline__ uint16_t rsqrtph_dev(uint16_t x, const uint16_t* tab)
{
    const uint32_t sign = x & 0x8000u;
    uint32_t m = x & 1023u;
    int E = (x >> 10) & 31;
    if (E == 31 && m) return (uint16_t)(x | 0x200u);
    if (E == 0 && m == 0) return (uint16_t)(sign | 0x7c00u);
    if (sign) return 0xfe00u;
    if (E == 31) return 0;
    if (E == 0) { const int lz = __clz((int)m) - 21; m = (m << lz) & 1023u; E = 1 - lz; }
    const int ue = E - 15, p = ue & 1, half = (ue - p) >> 1;
    const uint32_t t = tab[1024 + 1024 * p + m];
    return (uint16_t)(((((t >> 10) & 31u) - (uint32_t)half) << 10) | (t & 1023u));
}

// valid for every input whose reciprocal is not subnormal (always true for a VRSQRTPH result)
__device__ __forceinline__ uint16_t rcpph_dev(uint16_t x, const uint16_t* tab)
{
    const uint32_t sign = x & 0x8000u;
    uint32_t m = x & 1023u;
    int E = (x >> 10) & 31;
    if (E == 31) return (uint16_t)(m ? (x | 0x200u) : sign);
    if (E == 0) {
        if (m == 0) return (uint16_t)(sign | 0x7c00u);
        const int lz = __clz((int)m) - 21; m = (m << lz) & 1023u; E = 1 - lz;
    }
    const uint32_t t = tab[m];
    const int re = (int)((t >> 10) & 31u) + (15 - E);
    if (re >= 31) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | ((uint32_t)max(re, 0) << 10) | (t & 1023u));
}

// generic composition (any input); kept out of line, only NaN / infinity reach it from the hash
__device__ __attribute__((noinline)) uint16_t sqrt_ph_generic(uint16_t x, const uint16_t* tab)
{
    return rcpph_dev(rsqrtph_dev(x, tab), tab);
}

// VRCPPH(VRSQRTPH(v)), branch-free for everything finite (binary16 denormals are ordinary hash inputs: flat patches
// give radicands and eigenvalues below 6e-5):
//   positive (normal or denormal, normalised by a leading-zero count): y = VRSQRTPH(x) has the table mantissa
//            and exponent te - half, always normal (exponent 7..27); VRCPPH(y) = table row of y's mantissa with
//            exponent te2 + 15 - (te - half), in 2..23: none of the generic model's range checks can fire;
//   +-0 -> +-inf -> +-0;   negative (normal or denormal) -> QNaN 0xfe00 -> 0xfe00;
//   +-inf, NaN -> `rare`: the caller recomputes with the generic model.
__device__ __forceinline__ hf sqrt_ph(hf v, const uint16_t* ctab, bool& rare)
{
    // ctab = the two look-ups folded into one (device_abi.hip, composite_sqrt_table): C[p][m] = ((te2 + 15 - te) << 10) | mantissa of
    // VRCPPH's row for VRSQRTPH's mantissa, so that the result is C[p][m] + (half << 10) -- one LDS read instead of two dependent ones
    const uint32_t x = h_u(v);
    uint32_t m = x & 1023u;
    int E = (int)((x >> 10) & 31u);
    const int lz = __clz((int)(m | 1u)) - 21;                // m == 0 never takes the denormal values below
    const bool den = E == 0;
    m = den ? ((m << lz) & 1023u) : m;
    E = den ? 1 - lz : E;
    const int ue = E - 15, p = ue & 1, half = (ue - p) >> 1;
    uint32_t z = (uint32_t)((int)ctab[1024 * p + (int)m] + half * 1024);
    const bool zero = (x & 0x7fffu) == 0u;
    const bool negative = (x & 0x8000u) != 0u;
    z = negative ? 0xfe00u : z;
    z = zero ? x : z;
    rare |= ((x >> 10) & 31u) == 31u;
    return h_bits((uint16_t)z);
}

// GetHashValue_AVX512FP16_16h_{8,32}Elements (Raisr_AVX512FP16.cpp:382-471,497-590).  FAST: branch-free square roots,
// `rare` set when one of them saw an infinity or a NaN (the caller then calls the generic variant).
// thresholds BY VALUE: a reference into the kernel-argument struct would force it into scratch for the out-of-line variant
struct HashQ16 { uint16_t qangle, qs0, qs1, qc0, qc1, ls0, ls1; float cm0, cm1; int ct0, ct1, fold; };

// tab: FAST -> the composite table (in LDS); else the three instruction tables (global memory: only pixels that saw an infinity or a NaN)
template <bool FAST>
__device__ __forceinline__ int hash_px16_impl(hf a, hf b, hf d, const HashQ16 Q, const uint16_t* tab, bool& rare)
{
    auto root = [&](hf v) { return FAST ? sqrt_ph(v, tab, rare) : h_bits(rcpph_dev(rsqrtph_dev(h_u(v), tab), tab)); };
    const hf c100 = (hf)100.0f, one = (hf)1.0f;
    const hf pi = (hf)3.141592653f;
    const hf ONEQTR_PI = (hf)(3.14159265358979323846 / 4.0);
    const hf THRQTR_PI = (hf)(3.0 * 3.14159265358979323846 / 4.0);
    const hf k1963 = (hf)0.1963f, kn9817 = (hf)-0.9817f, tiny = (hf)1e-10f, near_zero = (hf)0.00000000000000001;
    a = a * c100; b = b * c100; d = d * c100;
    const hf T = a + d;
    const hf Dt = (a * d) - (b * b);
    // x / 4 and x / 2 are the same real numbers as x * 0.25 and x * 0.5, hence the same binary16 roundings
    // (also into the denormal range): no division needed for these two
    const hf rad = ((T * T) * (hf)0.25f) - Dt;
    const hf s = root(rad);
    const hf hT = T * (hf)0.5f;
    const hf L1 = hT + s, L2 = hT - s;
    const hf xx = (b < (hf)0.0f || b > (hf)0.0f) ? (L1 - d) : one;
    const hf ay = __builtin_fabsf16(b) + tiny;
    const bool neg = xx < (hf)0.0f;
    const hf xpa = xx + ay;
    const hf num = neg ? xpa : (xx - ay);
    const hf den = neg ? (ay - xx) : xpa;
    const hf rr = h_div(num, den);
    hf ang = neg ? THRQTR_PI : ONEQTR_PI;
    ang = __builtin_fmaf16(__builtin_fmaf16(k1963 * rr, rr, kn9817), rr, ang);
    const hf nang = (hf)-1.0f * ang;
    ang = (b < (hf)0.0f) ? nang : ang;
    ang = ang + ((ang < (hf)0.0f) ? pi : (hf)0.0f);
    const hf sL1 = root(L1), sL2 = root(L2);
    const float fl = __builtin_floorf((float)(ang * h_bits(Q.qangle)));
    int ai = (fl >= -32768.0f && fl <= 32767.0f) ? (int)fl : -32768;     // cvt_roundph_epi16, TO_NEG_INF
    ai = min(23, max(ai, 0));
    int si, ci;
    if (FAST && Q.fold) {
        // (finite operands only: the caller redoes pixels that saw an infinity or a NaN -- `rare` -- with the divisions)
        si = (int)(L1 >= h_bits(Q.ls0)) + (int)(L1 >= h_bits(Q.ls1));
        const float nf = (float)(sL1 - sL2), df = (float)((sL1 + sL2) + near_zero);
        const float b0 = Q.cm0 * df, b1 = Q.cm1 * df;                     // exact products
        const bool dpos = df > 0.0f;                                      // d == 0 means sL1 == sL2 == 0: 0 / 0, no threshold passes
        ci = (int)(dpos && (Q.ct0 ? nf >= b0 : nf > b0)) + (int)(dpos && (Q.ct1 ? nf >= b1 : nf > b1));
    } else {
        const hf coh = h_div(sL1 - sL2, (sL1 + sL2) + near_zero);
        const hf str = h_div(L1, c100);
        si = (int)(h_bits(Q.qs0) <= str) + (int)(h_bits(Q.qs1) <= str);
        ci = (int)(h_bits(Q.qc0) <= coh) + (int)(h_bits(Q.qc1) <= coh);
    }
    return ai * 9 + si * 3 + ci;
}

__device__ __attribute__((noinline)) int hash_px16_generic(hf a, hf b, hf d, const HashQ16 Q, const uint16_t* tab)
{
    bool unused = false;
    return hash_px16_impl<false>(a, b, d, Q, tab, unused);
}

// Test hook: the folded thresholds against the divisions they replace, EXHAUSTIVELY.  Strength: all 65 536 bit patterns of L1.
// Coherence: every pair (n, d) of binary16 bit patterns with d > 0 finite and n finite -- the pairs the fast hash can see; d is a
// sum of two VRCPPH(VRSQRTPH(.)) results, >= 0 or NaN, and d == 0 implies n == 0 -- plus NaN operands.  Counts disagreements.
__global__ __launch_bounds__(256) void k_debug_fold16(Pass16 Q, unsigned long long* __restrict__ out)
{
    const unsigned d_bits = blockIdx.x;                         // 0 .. 65535
    const hf dd = h_bits((uint16_t)d_bits);
    const hf c100 = (hf)100.0f;
    unsigned long long bad = 0, pairs = 0;
    const bool d_nan = (d_bits & 0x7c00u) == 0x7c00u && (d_bits & 0x3ffu);
    const bool d_ok = (d_bits > 0u && d_bits < 0x7c00u) || d_nan;          // positive finite, or NaN
    for (unsigned n_bits = threadIdx.x; n_bits < 65536u; n_bits += 256u) {
        const hf nn = h_bits((uint16_t)n_bits);
        if (d_bits == 0u) {                                     // this block also sweeps the strength comparison (n_bits plays L1)
            const hf str = h_div(nn, c100);
            const int s_div = (int)(h_bits(Q.qs0) <= str) + (int)(h_bits(Q.qs1) <= str);
            const int s_fold = (int)(nn >= h_bits(Q.ls0)) + (int)(nn >= h_bits(Q.ls1));
            bad += s_div != s_fold; pairs++;
        }
        const bool n_inf = (n_bits & 0x7fffu) == 0x7c00u;
        if (!d_ok || n_inf) continue;
        const hf coh = h_div(nn, dd);
        const int c_div = (int)(h_bits(Q.qc0) <= coh) + (int)(h_bits(Q.qc1) <= coh);
        const float nf = (float)nn, df = (float)dd;
        const float b0 = Q.cm0 * df, b1 = Q.cm1 * df;
        const bool dpos = df > 0.0f;
        const int c_fold = (int)(dpos && (Q.ct0 ? nf >= b0 : nf > b0)) + (int)(dpos && (Q.ct1 ? nf >= b1 : nf > b1));
        bad += c_div != c_fold; pairs++;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], pairs);
}

// ------------------------------------------------------------------------------------------------
// k_hash16: computeGTWG_Segment_AVX512FP16_16f (:138-224) + hash.  Same work shape as k_hash.
// Column accumulators per patch column, sequential over the 11 patch rows in binary16:
//   p = gx*w; A = fma(p,gx,A); B = fma(p,gy,B); q = gy*w; D = fma(q,gy,D)
// folded as (Gb+Gc)+(Ga+Gd) -- the association of sumitup2lane_AVX512FP16_16f (:67-75), which is
// the same for all four pixel classes up to operand order of single additions -- then scaled by NF
// in fp32 and rounded back (:197-221).
// ------------------------------------------------------------------------------------------------
// hash16_phase: one tile's tensor + hash once its LR window (origin (r0-6, c0-6), row stride LW) is in sL; returns the
// lane's R hashes (0xFF = pixel not filtered).
// Two pixel rows per packed register: the gradient tile holds VERTICAL pairs, entry (t, x) = {(gx[t], gx[t+1]), (gy[t], gy[t+1])}
// (8 bytes), so that rows j and j + 1 of a lane's four pixels share every instruction of a tap:
//   PX = X * W;  A2 = fma(PX, X, A2);  B2 = fma(PX, Y, B2);  PY = Y * W;  D2 = fma(PY, Y, D2)        (5 v_pk_*_f16 per 2 pixels)
// -- per pixel the operations and their order are those of the reference; only the packing differs (round 2 packed (gx, gy)
// of ONE pixel and spent a scalar v_fma_f16 plus a half-extract on B: 3.3 instructions per pixel and tap).
template <int R, int LW>
__device__ __forceinline__ void hash16_phase(const PassParams& P, const Pass16& Q, const GaussW16& gw, const hf* sL, uint2* sG,
                                             const uint16_t* sTab, int c0, int r0, unsigned (&hA)[R], unsigned long long* phase_ptr = nullptr)
{
#ifdef RAISR_HIP_DEV
#define RAISR_PHASE16(k) do { if (phase_ptr) phase_mark(*phase_ptr, (k), threadIdx.x); } while (0)
#else
#define RAISR_PHASE16(k)
#endif
    static_assert(R == 4, "two row pairs per lane");
    constexpr int TH = 4 * R;
    constexpr int GW_ = 74, GH = TH + 9;     // pair rows t = 0 .. TH + 8 (pair t covers gradient rows t and t + 1)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Gradient tile.  Pair row t holds the gradients of window rows t and t + 1: gx[t] = L[t+2][x+1] - L[t][x+1], gy[t] = L[t+1][x+2] - L[t+1][x],
    // and the second half of pair t is the first half of pair t + 1.  Lane = column, a wave walks down its 6-7 pair rows with the three
    // window rows of gx in registers: 3 LDS reads and 2 subtractions per row (round 6; before: every entry on its own -- 8 reads, 4
    // subtractions and an index division per entry, 58 reads per thread: 14.6 % of a wave's life, profiles/r06_call7_phase_cycles_C4.txt).
    // Same subtractions on the same operands.  The ten halo columns 64..73: one entry per thread, the old way.
    {
        const int wu = __builtin_amdgcn_readfirstlane(w);
        __builtin_assume(wu >= 0 && wu < 4);
        static_assert(GH == 25, "pair rows 0..24 over four waves: 7 + 6 + 6 + 6");
        const int t0 = wu == 0 ? 0 : 6 * wu + 1, cnt = wu == 0 ? 7 : 6;
        const hf* col = sL + t0 * LW + lane + 1;             // L[t0][x + 1]
        hf la = col[0], lb = col[LW];                        // L[t], L[t + 1]
        hf gxp = (hf)0.f, gyp = (hf)0.f;
#pragma unroll
        for (int i = 0; i <= 7; i++) {                       // single rows t0 + i, i = 0..cnt (cnt is wave-uniform)
            if (i <= cnt) {
                const hf lc = col[(i + 2) * LW];
                const hf gx = lc - la;
                const hf gy = col[(i + 1) * LW + 1] - col[(i + 1) * LW - 1];
                if (i > 0) sG[(t0 + i - 1) * GW_ + lane] = make_uint2(__builtin_bit_cast(uint32_t, (hf2){gxp, gx}), __builtin_bit_cast(uint32_t, (hf2){gyp, gy}));
                gxp = gx; gyp = gy; la = lb; lb = lc;
            }
        }
        if (threadIdx.x < (unsigned)(GH * (GW_ - 64))) {
            const int ty = (int)(threadIdx.x / (GW_ - 64)), tx = 64 + (int)(threadIdx.x - (unsigned)ty * (GW_ - 64));
            const hf* c = sL + ty * LW + tx + 1;             // column of the vertical differences
            const hf gx0 = c[2 * LW] - c[0], gx1 = c[3 * LW] - c[LW];
            const hf gy0 = c[LW + 1] - c[LW - 1], gy1 = c[2 * LW + 1] - c[2 * LW - 1];
            sG[ty * GW_ + tx] = make_uint2(__builtin_bit_cast(uint32_t, (hf2){gx0, gx1}), __builtin_bit_cast(uint32_t, (hf2){gy0, gy1}));
        }
    }
    __syncthreads();
    RAISR_PHASE16(1);                                        // gradient tile + barrier

    const hf2 z2 = {(hf)0.f, (hf)0.f};
    // The 11 patch columns in the reference's fold order, as four groups {0,8,4} {2,10,6} {1,9,5} {7,3} = Ga, Gd, Gb, Gc of
    // sumitup2lane: the fold bookkeeping runs once per group instead of once per column (a third of the rolled loop's
    // instructions were scalar selects), and a group's running sum starts from +0 (0 + S == S bit for bit: a column sum is
    // never -0 -- its fma chain starts from +0).  The next column's weights are fetched while the current column computes.
    hf2 hold_[3][2], t1_[3][2], cur_[3][2];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int p = 0; p < 2; p++) hold_[q][p] = t1_[q][p] = cur_[q][p] = z2;
    uint32_t wk[11], wn[11];
#pragma unroll
    for (int i = 0; i < 11; i++) wk[i] = gw.wT[c_col_order[0]][i];
    int kk = 0;
#pragma unroll 1
    for (int grp = 0; grp < 4; grp++) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int p = 0; p < 2; p++) cur_[q][p] = z2;
        const int len = grp == 3 ? 2 : 3;
#pragma unroll 1
        for (int j = 0; j < len; j++, kk++) {
            const int k = c_col_order[kk];
            const int kn = c_col_order[kk < 10 ? kk + 1 : 10];
#pragma unroll
            for (int i = 0; i < 11; i++) wn[i] = gw.wT[kn][i];
            uint2 g[13];
#pragma unroll
            for (int t = 0; t < 13; t++) g[t] = sG[(w * R + t) * GW_ + lane + k];
            hf2 A[2], B[2], D[2];
#pragma unroll
            for (int p = 0; p < 2; p++) A[p] = B[p] = D[p] = z2;
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const hf2 w2 = __builtin_bit_cast(hf2, wk[i]);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const hf2 X = __builtin_bit_cast(hf2, g[i + 2 * p].x), Y = __builtin_bit_cast(hf2, g[i + 2 * p].y);
                    const hf2 PX = X * w2;
                    A[p] = __builtin_elementwise_fma(PX, X, A[p]);
                    B[p] = __builtin_elementwise_fma(PX, Y, B[p]);
                    const hf2 PY = Y * w2;
                    D[p] = __builtin_elementwise_fma(PY, Y, D[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) { cur_[0][p] = cur_[0][p] + A[p]; cur_[1][p] = cur_[1][p] + B[p]; cur_[2][p] = cur_[2][p] + D[p]; }
#pragma unroll
            for (int i = 0; i < 11; i++) wk[i] = wn[i];
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (grp == 1) t1_[q][p] = hold_[q][p] + cur_[q][p];          // Ga + Gd
                if (grp == 0 || grp == 2) hold_[q][p] = cur_[q][p];          // Ga, later Gb
            }
    }
    RAISR_PHASE16(2);                                        // binary16 structure tensor (121 taps x 5 packed operations per pixel pair)
#undef RAISR_PHASE16
    hf2 *curA = cur_[0], *curB = cur_[1], *curD = cur_[2], *holdA = hold_[0], *holdB = hold_[1], *holdD = hold_[2];
    hf2 *t1A = t1_[0], *t1B = t1_[1], *t1D = t1_[2];

    const int c = c0 + lane;
    const HashQ16 HQ = {Q.qangle, Q.qs0, Q.qs1, Q.qc0, Q.qc1, Q.ls0, Q.ls1, Q.cm0, Q.cm1, Q.ct0, Q.ct1, Q.fold};
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const hf2 a2 = (holdA[p] + curA[p]) + t1A[p];
        const hf2 b2 = (holdB[p] + curB[p]) + t1B[p];
        const hf2 d2 = (holdD[p] + curD[p]) + t1D[p];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int j = 2 * p + e;
            const int r = r0 + w * R + j;
            // straight-line for every pixel of the lane (the four chains interleave); out-of-zone pixels are masked afterwards
            const hf a = h_scale_f32(e ? a2.y : a2.x, Q.nf), b = h_scale_f32(e ? b2.y : b2.x, Q.nf), d = h_scale_f32(e ? d2.y : d2.x, Q.nf);
            bool rare = false;
            unsigned h = (unsigned)hash_px16_impl<true>(a, b, d, HQ, sTab, rare);
            if (rare) h = (unsigned)hash_px16_generic(a, b, d, HQ, Q.tab16);
            hA[j] = (r < P.H - kMargin && c < P.c_final) ? h : 0xFFu;
        }
    }
}

template <int R, typename T>
__global__ __launch_bounds__(256, 4) void k_hash16(const T* __restrict__ lr, PassParams P, Pass16 Q, GaussW16 gw,
                                                    uint8_t* __restrict__ hash_out)
{
    constexpr int TH = 4 * R;
    constexpr int LW = 76, LH = TH + 12;
    __shared__ hf sL[LH * LW];
    __shared__ uint2 sG[(TH + 9) * 74];
    __shared__ uint16_t sTab[2048];          // composite square-root table (sqrt_ph)

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * 64, r0 = kMargin + by * TH;
    for (int i = threadIdx.x; i < 2048; i += 256) sTab[i] = Q.tab16[3072 + i];
    stage_tile<LH, LW, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, sL);   // u8 -> binary16 is exact for 8-bit content
    __syncthreads();
    unsigned hA[R];
    hash16_phase<R, LW>(P, Q, gw, sL, sG, sTab, c0, r0, hA);
    const int c = c0 + lane;
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int r = r0 + w * R + j;
        if (r < P.H - kMargin && c < P.c_final) hash_out[(size_t)r * P.hash_pitch + c] = (uint8_t)hA[j];
    }
}

// ------------------------------------------------------------------------------------------------
// k_filter16: DotProdPatch_AVX512FP16_16f (:227-242): 32 lanes x 4 chunks, then the tree of
// sumitup_AVX512FP16_16f (:77-109).  16 GPU lanes per pixel; lane l carries the reference's zmm
// lanes l and l+16 as a packed half2, so a chunk is one 4-byte filter load (the 16 lanes of a
// pixel read 64 contiguous bytes), two LDS reads and one v_pk_fma_f16:
//   r16[l] = a[l]+a[l+16] (inside the lane); r8[i]=r16[i]+r16[i+8]; t[i]=r8[i]+r8[i+4];
//   s0=t0+t2, s1=t1+t3; v=s0+s1      (DPP row rotations by 8, 4, 2, 1)
// HR plane is binary16 (u16 storage).
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ hf row_ror_h(hf v)
{
    const int bits = (int)h_u(v);
    return h_bits((uint16_t)__builtin_amdgcn_update_dpp(0, bits, CTRL, 0xf, 0xf, false));
}

// filter16_phase: one 64 x 16 tile once its LR window is in LDS (sL points at window position (r0-5, c0-5), row
// stride LW) and its hashes are in sH.
// PAIRS: the window comes as two arrays of packed pairs instead of single samples (build_pair_windows below), so that the two
// samples of a tap pair (k, k + 16) arrive with ONE ds_read_b32 instead of two ds_read_u16 and a v_perm to pack them.
template <int LW, bool PAIRS = false>
__device__ __forceinline__ void filter16_phase(const PassParams& P, const Pass16& Q, const hf* sL, const uint8_t* sH,
                                               int c0, int r0, uint16_t* __restrict__ hr, const uint32_t* sPA = nullptr, const uint32_t* sPB = nullptr)
{
    constexpr int TW = 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, l = lane & 15;
    int off0[4], off1[4];
    const uint32_t* pbase[4];                                          // PAIRS: array (A or B) and offset of the lane's pair per chunk
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const int k0 = 32 * ch + l, k1 = k0 + 16;
        off0[ch] = (k0 < kTaps) ? (k0 / 11) * LW + (k0 % 11) : 0;     // padding taps: coefficient is +0
        off1[ch] = (k1 < kTaps) ? (k1 / 11) * LW + (k1 % 11) : 0;
        // tap k0 + 16 sits one patch row down and 5 columns right of tap k0 (array A), or two rows down and 6 columns left (B)
        if (PAIRS) pbase[ch] = ((k0 % 11) + 5 < 11 ? sPA : sPB) + (k0 / 11) * LW + (k0 % 11);
    }
    const hf lo = (hf)P.lo, hi = (hf)P.hi;
    // bounds-checked 32-bit addressing of the binary16 bank: an unfiltered pixel's hash (0xFF) points past it and
    // loads +0, so its dot product is 0, fails the accept test (lo >= 0) and the pixel keeps LR -- no branch
    const __amdgpu_buffer_rsrc_t bank_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(Q.bank16), 0, Q.bank16_bytes, 0x00020000);
    const unsigned bank_stride = (unsigned)P.pixel_types * 256u;           // bytes per hash bucket
    const unsigned tcol = (P.pixel_types == 4) ? (unsigned)((g + 1) & 1) : 0u;   // (c-5)&1 with c = c0 + 4s + g, c0 even

#pragma unroll 1
    for (int row = 0; row < 4; row++) {
        const int prow = 4 * w + row;
        const int r = r0 + prow;
        unsigned keepb = 0u;                                               // binary16 bits of the kept pixel
        const unsigned trow = (P.pixel_types == 4) ? (unsigned)(((r - 5) & 1) * 2) : 0u;
        const unsigned row_lane_off = ((trow + tcol) * 64u + (unsigned)l) * 4u;
        // LDS byte addresses of the lane's 8 taps and of the centre pixel for step 0; step s adds the immediate 8*s
        const char* tp0[4];
        const char* tp1[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            tp0[ch] = reinterpret_cast<const char*>(sL + prow * LW + g + off0[ch]);
            tp1[ch] = reinterpret_cast<const char*>(sL + prow * LW + g + off1[ch]);
        }
        const char* ctr = PAIRS ? reinterpret_cast<const char*>(sPA + prow * LW + g + 5 * LW + 5)      // low half of a pair = the sample itself
                                : reinterpret_cast<const char*>(sL + prow * LW + g + 5 * LW + 5);
        const char* tpp[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++) tpp[ch] = reinterpret_cast<const char*>(PAIRS ? pbase[ch] + prow * LW + g : nullptr);
#define RAISR_LDS_H(p, s) (*reinterpret_cast<const hf*>((p) + (PAIRS ? 16 : 8) * (s)))
#define RAISR_LDS_PAIR(ch, s) (PAIRS ? __builtin_bit_cast(hf2, *reinterpret_cast<const uint32_t*>(tpp[ch] + 16 * (s))) \
                                     : (hf2){*reinterpret_cast<const hf*>(tp0[ch] + 8 * (s)), *reinterpret_cast<const hf*>(tp1[ch] + 8 * (s))})
        // one summation tree for the row's 16 steps, as in filter_phase: after every level two steps are merged into one register
        // and the next level runs once for both (the level-2 pairs stay inside their half of the row: row_shl:4 / row_shr:4);
        // lane l ends up with step sl = bitrev4(l)
        unsigned A16[16];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const unsigned hA = sH[prow * TW + 4 * s + g];
            const unsigned voff = __umul24(hA, bank_stride) + row_lane_off;
            hf2 acc = RAISR_LDS_PAIR(0, s) * __builtin_bit_cast(hf2, __builtin_amdgcn_raw_buffer_load_b32(bank_rsrc, voff, 0, 0));
#pragma unroll
            for (int ch = 1; ch < 4; ch++)
                acc = __builtin_elementwise_fma(RAISR_LDS_PAIR(ch, s),
                                                __builtin_bit_cast(hf2, __builtin_amdgcn_raw_buffer_load_b32(bank_rsrc, voff + 64u * ch, 0, 0)), acc);
            hf v = acc.x + acc.y;                           // a[l] + a[l+16]
            v = v + row_ror_h<0x128>(v);                    // r16[i] + r16[i+8]
            A16[s] = h_u(v);
        }
#define RAISR_MERGE(dst, src, mask) asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "s"(mask))
        unsigned B8[8], C4[4], D2[2];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned t = A16[2 * k];
            RAISR_MERGE(t, A16[2 * k + 1], 0xff00ff00ff00ff00ull);
            const hf x = h_bits((uint16_t)t);
            B8[k] = h_u((k & 1) ? x + row_ror_h<0x114>(x) : x + row_ror_h<0x104>(x));     // r8[i] + r8[i+4]
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            unsigned t = B8[2 * m];
            RAISR_MERGE(t, B8[2 * m + 1], 0xf0f0f0f0f0f0f0f0ull);
            const hf x = h_bits((uint16_t)t);
            C4[m] = h_u(x + row_ror_h<0x4e>(x));              // quad_perm [2,3,0,1]: t[i] + t[i+2]
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            unsigned t = C4[2 * n];
            RAISR_MERGE(t, C4[2 * n + 1], 0xccccccccccccccccull);
            const hf x = h_bits((uint16_t)t);
            D2[n] = h_u(x + row_ror_h<0xb1>(x));              // quad_perm [1,0,3,2]: s0 + s1
        }
        unsigned vb = D2[0];
        RAISR_MERGE(vb, D2[1], 0xaaaaaaaaaaaaaaaaull);
#undef RAISR_MERGE
        const hf v = h_bits((uint16_t)vb);
        const int sl = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);
        hf res = RAISR_LDS_H(ctr, sl);
        if (v > lo && v < hi) res = v;
        keepb = h_u(res);
#undef RAISR_LDS_H
#undef RAISR_LDS_PAIR
        const int c = c0 + 4 * sl + g;
        if (r < P.H - kMargin && c < P.c_final) hr[(size_t)r * P.hr_pitch + c] = (uint16_t)keepb;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_filter16(const T* __restrict__ lr, const uint8_t* __restrict__ hash,
                                                  PassParams P, Pass16 Q, uint16_t* __restrict__ hr)
{
    constexpr int TW = 64, TH = 16, LW = TW + 11, LH = TH + 10;
    __shared__ hf sL[LH * LW];
    __shared__ uint8_t sH[TH * TW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;

    stage_tile<LH, TW + 10, LW>(lr, P.lr_pitch, P.W, P.H, r0 - 5, c0 - 5, sL);
    for (int ty = w; ty < TH; ty += 4) {
        const int r = r0 + ty, c = c0 + lane;
        sH[ty * TW + lane] = (r < P.H - kMargin && c < P.c_final) ? hash[(size_t)r * P.hash_pitch + c] : (uint8_t)0xFFu;
    }
    __syncthreads();
    filter16_phase<LW>(P, Q, sL, sH, c0, r0, hr);
}

// Pair windows of the fused kernel's filter stage.  w = the 26 x 74 window of the tile (origin (r0-5, c0-5), inside the staged
// 28 x 77 window sW = sL + LW + 1).  A[y][x] = (w[y][x], w[y+1][x+5]), B[y][x] = (w[y][x], w[y+2][x-6]), row stride LW, as
// packed binary16 pairs.  Entries no tap pair ever reads are left out (A: x > 68; B: x < 6 or y = 25) -- their partner would lie
// outside the staged window.  Wave w builds rows w, w + 4, ...; lane = column (plus the ten columns 64..73 for lanes < 10).
template <int LW>
__device__ __forceinline__ void build_pair_windows(const hf* sW, uint32_t* sPA, uint32_t* sPB)
{
    const int lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wu >= 0 && wu < 4);
    auto one = [&](int y, int x) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(sW + y * LW + x);
        const uint32_t a = p[0];
        if (x <= 68) sPA[y * LW + x] = a | ((uint32_t)p[LW + 5] << 16);
        if (x >= 6 && y < 25) sPB[y * LW + x] = a | ((uint32_t)p[2 * LW - 6] << 16);
    };
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int y = wu + 4 * i;
        if (y < 26) {
            one(y, lane);
            if (lane < 10) one(y, 64 + lane);
        }
    }
}

// k_hashfilter16: both binary16 stages of a tile in one launch (see k_hashfilter).  The hash stage's gradient tile and tables and
// the filter stage's pair windows share one LDS region (a workgroup barrier on either side of build_pair_windows).
template <typename T>
__global__ __launch_bounds__(256, 6) void k_hashfilter16(const T* __restrict__ lr, PassParams P, Pass16 Q, GaussW16 gw,
                                                         uint8_t* __restrict__ hash_out, uint16_t* __restrict__ hr)
{
    constexpr int R = 4, TW = 64, TH = 16;
    constexpr int LW = 77, LH = TH + 12;
    constexpr int kGBytes = (TH + 9) * 74 * 8, kTabBytes = 2048 * 2, kPairBytes = 26 * LW * 4;
    // Array B starts 15 dwords after array A's end: with the arrays back to back (round 4), three of the four tap-pair chunks had
    // a lane on array A and a lane on array B of every 32-lane half hit the same LDS bank with different addresses (26 * 77 = 18
    // mod 32 puts B's columns 6..10 of one patch row on A's columns 0..5 of another): 6 extra LDS cycles on 8 per step, the bulk of
    // the kernel's SQ_LDS_BANK_CONFLICT (23.7 % of its LDS cycles).  Offsets 15 and 27 are the conflict-free ones at row stride 77
    // (exhaustive over strides 74..90 and offsets 0..39: tests/test_fp16_pair_windows.py replays the bank arithmetic).
    constexpr int kPairPad = 15 * 4;
    static_assert(2 * kPairBytes + kPairPad <= kGBytes + kTabBytes, "the pair windows fit the hash stage's region");
    __shared__ hf sL[LH * LW];
    __shared__ __attribute__((aligned(16))) unsigned char sRegion[kGBytes + kTabBytes];
    uint2* sG = reinterpret_cast<uint2*>(sRegion);
    uint16_t* sTab = reinterpret_cast<uint16_t*>(sRegion + kGBytes);
    uint32_t* sPA = reinterpret_cast<uint32_t*>(sRegion);
    uint32_t* sPB = reinterpret_cast<uint32_t*>(sRegion + kPairBytes + kPairPad);
    __shared__ uint8_t sH[TH * TW];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    const int c0 = kMargin + bx * TW, r0 = kMargin + by * TH;
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; hash_out += blockIdx.z * P.zs_hash;    // frame batches
    // the composite table (2048 binary16 entries): one 16-byte load per thread, in flight together with the window's loads.  (As a
    // loop of 2-byte loads the compiler issued five, waited, and then ran the last three one round trip each: five global round
    // trips in series at the head of every tile -- round 5, R5.10.)
    static_assert(kGBytes % 16 == 0, "the table's LDS copy is 16-byte aligned");
    const unsigned tid = threadIdx.x;
    RAISR_PHASE_DECL;                                        // development builds: wave-cycles per phase (scripts/phase_cycles.py C4)
    const uint4 tab_part = reinterpret_cast<const uint4*>(Q.tab16 + 3072)[threadIdx.x];
    {
        TileRegs<LH, 76, T> Rg;
        load_tile<LH, 76>(lr, P.lr_pitch, P.W, P.H, r0 - 6, c0 - 6, Rg);
        reinterpret_cast<uint4*>(sTab)[threadIdx.x] = tab_part;
        store_tile<LH, 76, LW>(Rg, sL);
    }
    RAISR_BARRIER(tid);
    RAISR_PHASE(0);                                          // window + table staging, barrier
    unsigned hA[R];
#ifdef RAISR_HIP_DEV
    hash16_phase<R, LW>(P, Q, gw, sL, sG, sTab, c0, r0, hA, &phase_t);
#else
    hash16_phase<R, LW>(P, Q, gw, sL, sG, sTab, c0, r0, hA);
#endif
    const int c = c0 + lane;
#pragma unroll
    for (int j = 0; j < R; j++) {
        sH[(w * R + j) * TW + lane] = (uint8_t)hA[j];
        const int r = r0 + w * R + j;
        if (P.write_hash && r < P.H - kMargin && c < P.c_final) hash_out[(size_t)r * P.hash_pitch + c] = (uint8_t)hA[j];
    }
    RAISR_PHASE(3);                                          // per-pixel hash (marks 1, 2 inside hash16_phase: gradient tile + barrier, tensor)
    RAISR_BARRIER(tid);                                      // every wave is done with the gradient tile and the tables
    build_pair_windows<LW>(sL + LW + 1, sPA, sPB);
    RAISR_BARRIER(tid);
    RAISR_PHASE(4);                                          // two barriers + pair windows
    filter16_phase<LW, true>(P, Q, sL + LW + 1, sH, c0, r0, hr, sPA, sPB);
    RAISR_PHASE(6);                                          // filter stage
}


// ------------------------------------------------------------------------------------------------
// k_blend16: CTCountOfBitsChangedSegment_AVX512FP16_16f (:258-355).  Columns below c_avx use the
// 32-wide binary16 body (:303-312), the rest the scalar fp32 tail (:326-352).
// ------------------------------------------------------------------------------------------------
template <typename TOut>
__global__ __launch_bounds__(256) void k_blend16(const TOut* __restrict__ lr, const uint16_t* __restrict__ hr,
                                                 PassParams P, Pass16 Q, TOut* __restrict__ out, int out_pitch)
{
    // Same shape as k_blend: tile 64 x 16, wave w owns rows [4w, 4w+4), lane = column; LR / HR tiles with a 1-px halo staged in
    // LDS with every load of a thread in flight before its first LDS write (wave = tile row, no index divisions), and a
    // 3 x 3 register window sliding down the wave's rows (6 LDS reads per pixel instead of 18).
    constexpr int TW = 64, TH = 16, LW = TW + 2, LH = TH + 2;
    __shared__ float sL[LH * LW];
    __shared__ float sHh[LH * LW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bx, by;
    xcd_tile(bx, by);
    lr += blockIdx.z * P.zs_lr; hr += blockIdx.z * P.zs_hr; out += blockIdx.z * P.zs_out;          // frame batches
    const int c0 = bx * TW, r0 = by * TH;
    {
        constexpr int NM = (LH + 3) / 4, REM = LW - 64;
        static_assert(LH * REM <= 256, "halo columns fit one sweep");
        const int wu = __builtin_amdgcn_readfirstlane(w);
        __builtin_assume(wu >= 0 && wu < 4);
        const int gxm = min(max(c0 - 1 + lane, 0), P.W - 1);
        const bool inxm = gxm >= kMargin && gxm < P.c_final;
        TOut lv[NM + 1];
        uint16_t hv[NM + 1];
        bool inz[NM + 1];
#pragma unroll
        for (int it = 0; it < NM; it++) {
            const int gy = min(max(r0 - 1 + min(wu + 4 * it, LH - 1), 0), P.H - 1);
            lv[it] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gxm];
            inz[it] = inxm && gy >= kMargin && gy < P.H - kMargin;
            hv[it] = inz[it] ? hr[(unsigned)gy * (unsigned)P.hr_pitch + (unsigned)gxm] : (uint16_t)0;
        }
        const int rty = min((int)(threadIdx.x / REM), LH - 1), rtx = 64 + (int)(threadIdx.x % REM);
        {
            const int gy = min(max(r0 - 1 + rty, 0), P.H - 1), gx = min(max(c0 - 1 + rtx, 0), P.W - 1);
            lv[NM] = lr[(unsigned)gy * (unsigned)P.lr_pitch + (unsigned)gx];
            inz[NM] = gy >= kMargin && gy < P.H - kMargin && gx >= kMargin && gx < P.c_final;
            hv[NM] = inz[NM] ? hr[(unsigned)gy * (unsigned)P.hr_pitch + (unsigned)gx] : (uint16_t)0;
        }
#pragma unroll
        for (int it = 0; it < NM; it++) {
            const int ty = wu + 4 * it;
            const float L = (float)lv[it];
            if (ty < LH) {
                sL[ty * LW + lane] = L;
                sHh[ty * LW + lane] = inz[it] ? (float)h_bits(hv[it]) : L;               // HR := LR outside the filtered zone
            }
        }
        if (threadIdx.x < LH * REM) {
            const float L = (float)lv[NM];
            sL[rty * LW + rtx] = L;
            sHh[rty * LW + rtx] = inz[NM] ? (float)h_bits(hv[NM]) : L;
        }
    }
    __syncthreads();
    const int x = c0 + lane;
    if (x >= P.W) return;
    float l[3][3], h[3][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            l[i + 1][j] = sL[(4 * w + i) * LW + lane + j];
            h[i + 1][j] = sHh[(4 * w + i) * LW + lane + j];
        }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int y = r0 + 4 * w + rr;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            l[0][j] = l[1][j]; l[1][j] = l[2][j]; h[0][j] = h[1][j]; h[1][j] = h[2][j];
            l[2][j] = sL[(4 * w + rr + 2) * LW + lane + j];
            h[2][j] = sHh[(4 * w + rr + 2) * LW + lane + j];
        }
        if (y >= P.H) break;
        const float Lc = l[1][1], Hc = h[1][1];
        int iv;
        if (x == 0 || y == 0 || x == P.W - 1 || y == P.H - 1) {
            iv = (int)Lc;
        } else {
            int hd = 0;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (i == 1 && j == 1) continue;
                    hd += ((l[i][j] < Lc) != (h[i][j] < Hc));
                }
            if (x < Q.c_avx) {
                const hf weight = (hf)((float)hd * 0.125f);          // hd / 8 exactly
                const hf w2 = (hf)1.0f - weight;
                hf val = (weight * (hf)Lc) + (w2 * (hf)Hc);
                val = val + (hf)0.5f;
                const float fl = __builtin_floorf((float)val);
                int fi = (fl >= -32768.0f && fl <= 32767.0f) ? (int)fl : -32768;
                if (fi < 0) fi = 0xFFFF;                              // cvtph_epu16 of a negative value
                iv = max(min(fi, P.ihi), P.ilo);
            } else {
                const float weight = (float)hd * 0.125f;
                float val = (weight * Lc) + ((1.0f - weight) * Hc);
                val = val + 0.5f;
                const float cl = val < P.lo ? P.lo : (val > P.hi ? P.hi : val);
                iv = (int)cl;
            }
        }
        out[(size_t)y * out_pitch + x] = (TOut)(iv << P.out_shift);
    }
}
