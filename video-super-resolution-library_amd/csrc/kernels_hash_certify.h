// kernels_hash_certify.h -- certified hashing: separable approximate tensor, approx_hash with its error bounds, exact 16-lane fallback, hash_phase_ac
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

// ------------------------------------------------------------------------------------------------
// Certified hashing ("approximate, then certify").  The output needs the exact BUCKET, not the exact
// tensor.  k_hashfilter_ac therefore computes the structure tensor cheaply -- the literal 11 x 11 table is a rank-1
// table up to its 6-digit truncation, so a separable 11 + 11-tap pass over the per-pixel gradient products
// (gx^2, gx gy, gy^2) reproduces (a, b, d) to a relative eps of a few 1e-6 -- evaluates the hash quantities in plain
// fp32 (native sqrt / rcp), and accepts the bucket only when rigorous bounds on
//     |quantity the reference's instruction sequence produces  -  quantity computed here|
// keep angle*24/pi, strength and coherence away from every bucket boundary and every discontinuity of the
// reference's code (sign of b, xx, L2, the "ang < 0" wrap).  Everything else -- a few per cent of the pixels on
// noisy content, 1-D structures and exactly symmetric patterns on synthetic content -- goes through a per-tile
// worklist to the exact code (exact_pixel below: the same operations in the same order as hash_phase).
// The bounds (tests/tools/certify_proto.py derives and validates them against the oracle):
//   tensor      |a-a'| <= eps a', |d-d'| <= eps d', |b-b'| <= eps T'/2,  T' = a'+d'
//               eps = 1.05 (eps_w + 48 u): eps_w = max |us_i us_k / w_ik - 1| over the ACTUAL fp32 constants of both paths,
//               16 u for the reference's 12-term fma chains + 4-level fold, 23 u for the separable passes (u = 2^-24).
//   root        reference: rad = fl(T^2/4 - (ad - b^2)) = R + eta, R = ((a-d)/2)^2 + b^2, |eta| <= 2 u T^2;  s = sqrt14(rad),
//               sqrt14(x) = sqrt(x)(1 + e), |e| <= E (1.0e-4 for VRCP14(VRSQRT14), 6.5e-4 for RCPPS(RSQRTPS); enumerated in
//               tests/test_certify_bounds.py).  With s* = sqrt(R'):   |s - s*| <= E_s = 1.42 eps T' + (2e-7 + eps^2) T'^2 / s* + 1.05 E s*
//   L1, L2, xx  |. - .*| <= E_L = E_s + (eps/2 + 6 u) T'   (L2* = (a'd' - b'^2) / L1* carries 4 u T' more)
//   coherence   t = sqrt(L2/L1), rho = 0.55 (E_L2/(L2*-E_L2) + E_L/(L1*-E_L)) + 2.4 E: |coh - coh*| <= 2.07 t* rho / (1 + t*)^2 + 2e-6
//               for rho <= 1/16, <= 2 t* rho + 2e-6 otherwise; needs L2* > 2 E_L2
//   angle       rr = (xx - ay)/(xx + ay), |d rr| <= 2((ay+E_ay) E_L + (xx+E_L) E_ay) / (xx + ay - E_L - E_ay)^2 + 4 u,
//               |P'(rr)| < 1 for the cubic P  =>  |ang_raw - ang_raw*| <= d rr + 1.5e-6;  q = ang 24/pi: + 2e-5
// ------------------------------------------------------------------------------------------------
struct SepW {
    float us[11];                // separable weights, sqrt(NF) folded in: us[i] us[k] ~ wT[k][i]
    float es1, es2;              // 1.42 eps, 2e-7 + eps^2
    float eEL, eEb;              // eps/2 + 6 u, eps/2
    float e105[2], e24[2];       // 1.05 E and 2.4 E for the AVX-512 (0) and the AVX2 (1) approximation instructions
    float c1e;                   // 1.05 eps: half-width of the box around a' that contains the reference's a (one-dimensional windows, class1_hash)
};

struct HashQf { float qangle, qs0, qs1, qc0, qc1; };

// returns true when the bucket is certified
__device__ __forceinline__ bool approx_hash(float a, float b, float d, const HashQf Q, const SepW& S, int fl, unsigned& bucket,
                                            int* si_out = nullptr, bool* ok_str_out = nullptr)
{
    const float U1 = 5.9604645e-8f;                      // 2^-24
    const float pi = 3.141592653f;
    const float ONEQTR_PI = (float)(3.14159265358979323846 / 4.0);
    const float T = a + d;
    const float m = 0.5f * (a - d);
    const float bb = b * b;
    const float R = __builtin_fmaf(m, m, bb);
    const float s = __builtin_amdgcn_sqrtf(R);
    const float hT = 0.5f * T;
    const float L1 = hT + s;
    const float rL1 = __builtin_amdgcn_rcpf(L1);
    const float det = __builtin_fmaf(a, d, -bb);
    const float L2 = det * rL1;                          // = T/2 - s without the cancellation
    const float rs = __builtin_amdgcn_rcpf(s);
    const float E_s = __builtin_fmaf(S.es1, T, __builtin_fmaf((S.es2 * T) * T, rs, S.e105[fl] * s));
    bool ok = (T > 0.0f) & (s > 0.0f) & (E_s <= 0.25f * s);
    const float E_L = __builtin_fmaf(S.eEL, T, E_s);
    const float E_L2 = __builtin_fmaf(4.0f * U1, T, E_L);
    // strength
    ok &= (__builtin_fabsf(L1 - Q.qs0) > E_L) & (__builtin_fabsf(L1 - Q.qs1) > E_L);
    const int si = (int)(Q.qs0 <= L1) + (int)(Q.qs1 <= L1);
    if (si_out) *si_out = si;
    if (ok_str_out) *ok_str_out = ok;                    // preconditions of the root bound + strength index certified
    // coherence
    ok &= L2 > 2.0f * E_L2;
    const float t = __builtin_amdgcn_sqrtf(L2 * rL1);
    const float r1t = __builtin_amdgcn_rcpf(1.0f + t);
    const float coh = (1.0f - t) * r1t;
    // rho: relative bound on |t_ref - t|; d coh / dt = -2 / (1 + t)^2, and with rho <= 1/16 the reference's 1 + t_ref is within
    // 1/32 of 1 + t: 2 / ((1 + t_ref)(1 + t)) <= 2.0646 / (1 + t)^2 (round 5: the factor 1 / (1 + t)^2 -- 0.39 at the threshold
    // coh = 0.41, 0.46 at 0.19 -- was bounded by 1, which sent twice as many pixels to the exact path for their coherence)
    const float rho = __builtin_fmaf(0.55f, __builtin_fmaf(E_L2, __builtin_amdgcn_rcpf(L2 - E_L2), E_L * __builtin_amdgcn_rcpf(L1 - E_L)), S.e24[fl]);
    // (rho > 1/16 -- 1-D structures, whose L2 is of the size of its own bound: the denominators are only known to be >= 1)
    // (the 2.07 / (1 + t)^2 form is derived for t <= 1; t > 1 cannot pass the L2 test above with L2 <= L1, the guard states it in code)
    const float slope = ((rho <= 0.0625f) & (t <= 1.0f)) ? 2.07f * (r1t * r1t) : 2.0f;
    const float dcoh = __builtin_fmaf(slope * t, rho, 2e-6f);
    ok &= (__builtin_fabsf(coh - Q.qc0) > dcoh) & (__builtin_fabsf(coh - Q.qc1) > dcoh);
    const int ci = (int)(Q.qc0 <= coh) + (int)(Q.qc1 <= coh);
    // angle
    const float E_b = S.eEb * T;
    const float ab = __builtin_fabsf(b);
    const float ay = ab + 1e-10f;
    const float xx = m >= 0.0f ? m + s : bb * __builtin_amdgcn_rcpf(s - m);        // (s - m)(s + m) = b^2
    const float E_ay = __builtin_fmaf(2.0f * U1, ay, E_b);        // |b| vs |b'|, and the rounding of "+ 1e-10" on EITHER side (round 5: was 1 u, covered
                                                                    // only by slack elsewhere -- found by tests/test_certify_interval.py)
    const float xpa = xx + ay;
    const float D = xpa - (E_L + E_ay);
    const float rr = (xx - ay) * __builtin_amdgcn_rcpf(xpa);
    const float rD = __builtin_amdgcn_rcpf(D);
    const float drr = __builtin_fmaf(2.0f * __builtin_fmaf(ay + E_ay, E_L, (xx + E_L) * E_ay), rD * rD, 4.0f * U1);
    const float ang_raw = __builtin_fmaf(__builtin_fmaf(0.1963f * rr, rr, -0.9817f), rr, ONEQTR_PI);
    const float dang = drr + 1.5e-6f;
    float ang = b < 0.0f ? -ang_raw : ang_raw;
    ang = ang < 0.0f ? ang + pi : ang;
    const float q = ang * Q.qangle;
    const float dq = __builtin_fmaf(Q.qangle, dang, 2e-5f);
    const float k = __builtin_fminf(__builtin_fmaxf(__builtin_floorf(q), 0.0f), 23.0f);
    const float fr = q - k;
    const bool c_ang = (xx > 2.0f * E_L) & (D > 0.0f) & (ab > E_b) & (__builtin_fabsf(ang_raw) > dang) &
                       ((k < 1.0f) | (fr > dq)) & ((k > 22.0f) | (1.0f - fr > dq));
    ok &= c_ang;                                         // (a window without any gx or gy has L2 = 0 and is never certified)
    bucket = (unsigned)((int)k * 9 + si * 3 + ci);
    return ok;
}

// ------------------------------------------------------------------------------------------------
// Exactly ONE-DIMENSIONAL windows ("class 1", round 6; docs/CERTIFY.md s9).  Real pictures -- above all compressed ones: 8 x 8 block
// edges over flat ground, letterbox borders, flat graphics -- are full of windows in which every gy (or every gx) is 0.  Their
// approximate tensor is (a', 0, 0) [or (0, 0, d')] with EXACT zeros (sums of zeros; nothing underflows, docs/CERTIFY.md s1), and then
// the reference's tensor is (a, +-0, 0) exactly as well, |a - a'| <= eps a'.  No bound certifies such a pixel (b = 0, L2 ~ 0, xx
// undefined: three discontinuities at once) -- on JPEG-like content they were 60-70 % of the uncertified pixels and overflowed the
// worklist of 12-44 % of the tiles.  But the reference's hash of (a, 0, 0) is almost a constant:
//   angle      b == 0 -> xx = 1, atan2Approximation(1e-10, 1) = a constant: the angle index of the zero tensor's bucket;
//   strength   L1 = a/2 + sqrt14(a^2/4): the bound E_L of approx_hash applies unchanged (si, ok_str);
//   coherence  L2 = fl(a/2 - sqrt14(fl(a a)/4)) is the table error of VRCP14(VRSQRT14(.)) at a^2/4, NOT a small number with a sign one
//              could bound: L2 >= 0 -> sqrt(L2/L1) <= 0.0071 (AVX2 flavour: 0.0181) -> coh >= 0.985 (0.964) -> index 2;  L2 < 0 -> the
//              reference's sqrt14(L2) is NaN -> coh NaN -> index 0 in the AVX-512 flavour ([Q <= NaN] = 0), index 2 in the AVX2 flavour
//              (2 - [NaN <= Q0] - [NaN <= Q1]).  The AVX2 flavour's index is therefore 2 whatever the sign.
// sign(L2) is a function of the MANTISSA of a alone (a -> 4a scales every quantity exactly by 4 or 2), so it is tabulated: c1tab[i], i =
// mantissa >> 7 (65 536 buckets of 128 consecutive floats), bit 0 = some a of the bucket has L2 < 0, bit 1 = some has L2 >= 0, built once
// per context by k_build_c1tab with the exact models of x86_approx_dev.h.  The pixel is certified when every bucket the box
// [a'(1 - 1.05 eps), a'(1 + 1.05 eps)] touches (at most three) carries the same single bit -- 72 % of the class on uniformly distributed
// mantissas (tests/test_class1.py enumerates all 2^23 mantissas against the oracle's hash and replays the bucket logic).
// Precondition per model (device_abi.hip, configure): both coherence thresholds below 0.98 (AVX2 flavour: 0.96).
// ------------------------------------------------------------------------------------------------
constexpr unsigned kC1Buckets = 65536u;
__global__ __launch_bounds__(256) void k_build_c1tab(const uint2* __restrict__ tab14, uint8_t* __restrict__ out)
{
    __shared__ uint2 sT[128];
    if (threadIdx.x < 128) sT[threadIdx.x] = tab14[threadIdx.x];
    __syncthreads();
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    unsigned flags = 0u;
    for (unsigned k = 0; k < 128u; k++) {
        // the reference's own operations on the tensor (a, 0, 0) (Raisr_AVX512.cpp:185-202), a in [1, 2)
        const float a = __uint_as_float(0x3f800000u | (i << 7) | k), b = 0.0f, d = 0.0f;
        const float T = a + d;
        const float Dt = (a * d) - (b * b);
        const float rad = ((T * T) / 4.0f) - Dt;
        const float s = x86dev::rcp14(x86dev::rsqrt14(rad, sT + 64), sT);
        const float L2 = (T / 2.0f) - s;
        flags |= (L2 < 0.0f) ? 1u : 2u;
    }
    out[i] = (uint8_t)flags;
}

// class-1 decision of one pixel: a1 = the non-zero diagonal entry of its approximate tensor.  Returns whether the AVX-512 flavour's
// coherence index is certain and writes it to ci0 (the AVX2 flavour's is 2).
__device__ __forceinline__ bool class1_coherence(float a1, const SepW& S, const uint8_t* __restrict__ c1tab, unsigned& ci0)
{
    const float e = S.c1e * a1;
    const unsigned lo = __float_as_uint(a1 - e), hi = __float_as_uint(a1 + e);
    const unsigned il = (lo >> 7) & (kC1Buckets - 1u), ih = (hi >> 7) & (kC1Buckets - 1u);
    const unsigned span = (ih - il) & (kC1Buckets - 1u);                  // cyclic: the box may straddle a power of two (mantissa wraps)
    const unsigned im = span >= 2u ? (il + 1u) & (kC1Buckets - 1u) : il;
    const unsigned t = (unsigned)c1tab[il] | (unsigned)c1tab[im] | (unsigned)c1tab[ih];
    ci0 = t == 1u ? 0u : 2u;
    // magnitudes for which a^2 / 4 and L2 are normal numbers and the scale invariance holds (content: 4.5e-15 <= a <~ 1)
    return (span <= 2u) & ((t == 1u) | (t == 2u)) & (a1 > 1e-17f) & (a1 < 1e17f);
}

// first (bA) / second (bB) hash of a class-1 pixel in a column that hashes with the AVX-512 flavour (inA) and / or the AVX2 flavour (inB);
// si / ok_str: strength index and "preconditions + strength certified" of approx_hash.  Returns whether the pixel is certified.
__device__ __forceinline__ bool class1_buckets(float a1, unsigned si, bool ok_str, const SepW& S, const PassParams& P, bool inA, bool inB,
                                               unsigned& bA, unsigned& bB)
{
    unsigned ci0;
    const bool ci_ok = class1_coherence(a1, S, P.c1tab, ci0);
    const unsigned b0 = (unsigned)(P.zero_bucket[0] / 9) * 9u + si * 3u + ci0;      // AVX-512 flavour: the zero tensor's angle index (b == 0)
    const unsigned b1 = (unsigned)(P.zero_bucket[1] / 9) * 9u + si * 3u + 2u;       // AVX2 flavour
    const bool okA = inA ? (ci_ok && (P.c1_ok & 1)) : true;                          // the flavours this column hashes with
    const bool okB = inB ? (P.c1_ok & 2) != 0 : true;
    bA = inA ? b0 : b1;
    bB = b1;
    return ok_str && okA && okB;
}

// first / second hash of a pixel in column c from its exact tensor: the flavour logic of hash_phase's epilogue
__device__ __forceinline__ void flavour_hash(const PassParams& P, const uint2* sTab, float a, float b, float d, int c, unsigned& hA, unsigned& hB)
{
    const HashQ HQ = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1, P.lut_legacy};
    const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
    hA = 0xFFu; hB = 0xFFu;
    unsigned h = 0xFFu;
    if (inA) {
        bool rare = false;
        h = (unsigned)hash_px_impl<0>(a, b, d, HQ, sTab, rare);
        if (rare) h = (unsigned)hash_px_generic(a, b, d, HQ, sTab);
    }
    if (inB) {
        const unsigned hL = (unsigned)hash_px_legacy(a, b, d, HQ, sTab);
        if (inA) hB = hL; else h = hL;
    }
    hA = (inA || inB) ? h : 0xFFu;
}

// Exact tensor of up to four worklist pixels per wave with 16 lanes per pixel: lane l < 11 of a group runs the
// reference's chain of patch column l (the 11 patch rows in order, weights wl[i] = wT[l][i]); the 11 column sums are
// folded in sumitup_ps_512's association with DPP row shifts (row_shl:n -- lane i reads lane i+n of its row of 16):
//   u = S + shl8(S): lanes 0..2 = S0+S8, S1+S9, S2+S10;  v = u + shl4(S): Ga, Gb, Gd;  lane 3: S3 + S7 = Gc;
//   r = z + shl2(z): lane 0 = Ga+Gd, lane 1 = Gb+Gc;  lane 0: (Ga+Gd) + (Gb+Gc)  [fp addition commutes bit for bit].
// The latency of one round is ~11 rows instead of 121 taps: what matters when only a handful of pixels per tile need it.
template <int CTRL>
__device__ __forceinline__ float row_shl(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float fold11(float S, bool lane3)
{
    const float u = S + row_shl<0x108>(S);
    const float v = u + row_shl<0x104>(S);
    const float wv = S + row_shl<0x104>(S);
    const float z = lane3 ? wv : v;
    const float r = z + row_shl<0x102>(z);
    return r + row_shl<0x101>(r);
}
template <typename GT>
__device__ __forceinline__ void exact_tensor16(const GT* sG, const float (&wl)[11], int prow, int pcol, int l, float& a, float& b, float& d)
{
    constexpr int GW_ = 74;
    const GT* base = sG + prow * GW_ + pcol + min(l, 10);
    f2 AD = {0.f, 0.f};
    float B = 0.f;
    f2 gg[11];
#pragma unroll
    for (int i = 0; i < 11; i++) gg[i] = grad_load(base + i * GW_);
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const f2 w2 = {wl[i], wl[i]};
        const f2 pq = gg[i] * w2;
        AD = __builtin_elementwise_fma(pq, gg[i], AD);
        B = __builtin_fmaf(pq.x, gg[i].y, B);
    }
    const bool lane3 = l == 3;
    a = fold11(AD.x, lane3);
    b = fold11(B, lane3);
    d = fold11(AD.y, lane3);
}

// Approximate structure tensor of the lane's RPW pixels (rows [RPW w, RPW w + RPW) of the tile, column = lane): separable 11 + 11
// taps on the gradient products.  V pass: lane-task (x, rg) = column x of the gradient tile, output rows [RPW rg, RPW rg + RPW),
// results as one RPW-vector per (channel, row group, column) in sV; workgroup barrier; H pass: lane = column, wave = row group.
// (RPW = 4: the 64 x 16 tile; RPW = 2: the 64 x 8 tile.)
template <int N> struct FVec { typedef float type __attribute__((ext_vector_type(N))); };

template <int RPW, typename GT>
__device__ __forceinline__ void tensor_acN(const SepW& S, const GT* sG, typename FVec<RPW>::type* sV, float (&ta)[RPW], float (&tb)[RPW], float (&td)[RPW],
                                           unsigned tid = threadIdx.x)
{
    using fv = typename FVec<RPW>::type;
    constexpr int GW_ = 74;
    const int lane = tid & 63, w = tid >> 6;
    auto vpass = [&](int x, int rg) {
        float va[RPW], vb[RPW], vd[RPW];
#pragma unroll
        for (int r = 0; r < RPW; r++) { va[r] = 0.f; vb[r] = 0.f; vd[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < RPW + 10; t++) {
            float pa, pb, pd;
            grad_products(sG + (RPW * rg + t) * GW_ + x, pa, pb, pd);
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int i = t - r;
                if (i >= 0 && i < 11) {
                    va[r] = __builtin_fmaf(S.us[i], pa, va[r]);
                    vb[r] = __builtin_fmaf(S.us[i], pb, vb[r]);
                    vd[r] = __builtin_fmaf(S.us[i], pd, vd[r]);
                }
            }
        }
        fv oa, ob, od;
#pragma unroll
        for (int r = 0; r < RPW; r++) { oa[r] = va[r]; ob[r] = vb[r]; od[r] = vd[r]; }
        sV[(0 * 4 + rg) * GW_ + x] = oa;
        sV[(1 * 4 + rg) * GW_ + x] = ob;
        sV[(2 * 4 + rg) * GW_ + x] = od;
    };
    vpass(lane, w);
    if (w == 0 && lane < 40) vpass(64 + lane % 10, lane / 10);       // the 10 halo columns of all four row groups
    RAISR_BARRIER(tid);
#pragma unroll
    for (int r = 0; r < RPW; r++) { ta[r] = 0.f; tb[r] = 0.f; td[r] = 0.f; }
    // software-pipelined by hand (next column's three vectors in flight during this column's FMAs) and fenced per
    // column: left alone, the scheduler issues all 33 loads first and sinks the FMAs into the hash code -- 132 live VGPRs
    fv xa = sV[(0 * 4 + w) * GW_ + lane], xb = sV[(1 * 4 + w) * GW_ + lane], xd = sV[(2 * 4 + w) * GW_ + lane];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        fv na = xa, nb = xb, nd = xd;
        if (k < 10) {
            na = sV[(0 * 4 + w) * GW_ + lane + k + 1];
            nb = sV[(1 * 4 + w) * GW_ + lane + k + 1];
            nd = sV[(2 * 4 + w) * GW_ + lane + k + 1];
        }
        const float uk = S.us[k];
#pragma unroll
        for (int r = 0; r < RPW; r++) ta[r] = __builtin_fmaf(uk, xa[r], ta[r]);
#pragma unroll
        for (int r = 0; r < RPW; r++) tb[r] = __builtin_fmaf(uk, xb[r], tb[r]);
#pragma unroll
        for (int r = 0; r < RPW; r++) td[r] = __builtin_fmaf(uk, xd[r], td[r]);
        xa = na; xb = nb; xd = nd;
        __builtin_amdgcn_sched_barrier(0);
    }
    // pin the sums here: otherwise LLVM sinks the FMAs of rows 1.. into the per-pixel hash code and keeps (spills) the loaded columns
#pragma unroll
    for (int r = 0; r < RPW; r++) asm volatile("" : "+v"(ta[r]), "+v"(tb[r]), "+v"(td[r]));
}

template <typename GT>
__device__ __forceinline__ void tensor_ac(const SepW& S, const GT* sG, float4* sV, float (&ta)[4], float (&tb)[4], float (&td)[4], unsigned tid = threadIdx.x)
{
    tensor_acN<4, GT>(S, sG, reinterpret_cast<typename FVec<4>::type*>(sV), ta, tb, td, tid);
}

// In-tile worklist of k_hashfilter_ac: more uncertain pixels than this and the whole tile takes the all-exact routine -- it then
// pays the approximate AND the exact work.  160 entries: measured 48 / 96 / 160 / 256 / 320 (scripts/r03_call28.sh, r03_call29.sh);
// with 48, tiles of AVX2-flavour frames (twice the uncertified share: their table error is 6.5e-4) overflowed often enough that C1
// ran 13 % (random frames: 21 %) slower than with 160, C2 / C3 / C5 1-2 %; beyond 160 nothing changes.
// Round 6, on photographs (whose lists are longer: 1.2-8 % uncertified against 0.4 % on the synthetic frame, 3 % of the tiles past 160):
// 256 entries -- all that fits (tensors behind the table in sV's space, one lane per entry in the hash round) -- C2 +2.4 %, C1 +4.4 %,
// C2b +2.1 % on the photo frames, synthetic kinds unchanged (docs/EXPERIMENTS.md R6.11).
#ifdef RAISR_EXP_OCC5
constexpr unsigned kListMax = 88;
#else
constexpr unsigned kListMax = 256;
#endif
static_assert(kListMax <= 256, "the hash round of the worklist runs one lane per entry");

// hash stage of k_hashfilter_ac for one tile: sG holds the gradient tile.  Leaves the buckets of the tile in sH / sH2
// (0xFF: not filtered / no re-hash) and ends with a workgroup barrier.
template <int LW, typename GT, int RPW = 4, bool PC = false>
__device__ __forceinline__ void hash_phase_ac(const PassParams& P, const GaussW& gw, const SepW& S, const float* sL, GT* sG, typename FVec<RPW>::type* sV,
                                              const uint2* sTab, uint8_t* sH, uint8_t* sH2, uint16_t* sList, unsigned* sCnt,
                                              int c0, int r0, unsigned tid = threadIdx.x, float* sQ = nullptr)
{
    constexpr int GW_ = 74, TW = 64;
    const int lane = tid & 63, w = tid >> 6;
    if (tid == 0) sCnt[0] = 0;
    float ta[RPW], tb[RPW], td[RPW];
    RAISR_PHASE_DECL;
    tensor_acN<RPW, GT>(S, sG, sV, ta, tb, td, tid);
    RAISR_PHASE(2);                                        // V pass + barrier + H pass
    // ---- approximate hash + certification of the lane's 4 pixels ----
    const int c = c0 + lane;
    const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
    const int fl = inB ? 1 : 0;                            // the AVX2 flavour's wider table error covers the re-hashed columns too
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    unsigned nUnc = 0, certbits = 0;
    unsigned bkA[RPW], bkB[RPW];                           // first / second hash candidates of the lane's pixels
    unsigned c1bits = 0, okbits = 0, sibits = 0;
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        unsigned bucket;
        int si;
        bool ok_str;
        bool cert = approx_hash(ta[j], tb[j], td[j], Q, S, fl, bucket, &si, &ok_str);
        const bool zero = (ta[j] + td[j]) == 0.0f;          // flat window: the reference's tensor is exactly (0, 0, 0) as well
        cert |= zero;
        // first hash: AVX-512 flavour where the column has one, else the AVX2 flavour; second hash: AVX2 flavour of the
        // re-hashed columns.  A certified bucket holds for both flavours (fl selects the wider table error there); the
        // zero tensor's bucket is looked up per flavour (computed once per tile with the exact code).
        bkA[j] = zero ? (unsigned)P.zero_bucket[inA ? 0 : 1] : bucket;
        bkB[j] = zero ? (unsigned)P.zero_bucket[1] : bucket;
        certbits |= (cert ? 1u : 0u) << j;
        // exactly one-dimensional window (class 1 above): decided below, outside this straight-line code
        const bool c1 = !zero && tb[j] == 0.0f && (ta[j] == 0.0f || td[j] == 0.0f);
        c1bits |= (c1 ? 1u : 0u) << j;
        okbits |= (ok_str ? 1u : 0u) << j;
        sibits |= (unsigned)si << (2 * j);
    }
    if (P.c1tab && __any(c1bits != 0u)) {                  // wave-uniform: natural content rarely enters
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            if ((c1bits >> j) & 1u) {
                unsigned bA, bB;
                if (class1_buckets(fmaxf(ta[j], td[j]), (sibits >> (2 * j)) & 3u, (okbits >> j) & 1u, S, P, inA, inB, bA, bB)) {
                    bkA[j] = bA;
                    bkB[j] = bB;
                    certbits |= 1u << j;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        const int prow = RPW * w + j;
        const int r = r0 + prow;
        const bool zone = r < P.H - kMargin && c < P.c_final && (inA || inB);
        const bool cert = (certbits >> j) & 1u;
        const bool unc = zone && (!cert || P.cert_check);
        sH[prow * TW + lane] = zone ? (uint8_t)bkA[j] : (uint8_t)0xFFu;
        sH2[prow * TW + lane] = (zone && inA && inB) ? (uint8_t)bkB[j] : (uint8_t)0xFFu;
        if (unc) {
            const unsigned slot = atomicAdd(sCnt, 1u);
            if (slot < kListMax) sList[slot] = (uint16_t)((prow << 6) | lane | (cert ? 0x8000 : 0));
#ifdef RAISR_PROBE_L2CERT                                  /* TIMING PROBE (scripts/build_exp.sh l2probe -DRAISR_PROBE_L2CERT), output wrong for ~0.02 % of the pixels: */
            if (slot < kListMax) reinterpret_cast<float4*>(sG)[slot] = float4{ta[j], tb[j], td[j], 0.0f};    // the listed pixel's APPROXIMATE tensor (the gradient tile is dead: no exact tensor follows)
#endif
        }
        nUnc += (zone && !cert) ? 1u : 0u;
    }
    if (P.cert_stats) {
        if (nUnc) atomicAdd(&sCnt[1], nUnc);
    }
    RAISR_PHASE(3);                                        // approximate hash + certification
    RAISR_BARRIER(tid);
    RAISR_PHASE(4);                                        // ... wait for the other waves
    // pair-column filter stage: its second window copy goes into sV's space now that every wave is past its H pass (behind the table
    // and the tensors of the exact path); the barriers of the worklist -- or the one at the end -- publish it
    if constexpr (PC) {
#pragma unroll
        for (unsigned i = tid; i < 26u * LW; i += 256u) sQ[i] = sL[LW + 1 + i];
    }

    // ---- worklist: the exact path for what could not be certified ----
    const unsigned n = sCnt[0];
    unsigned bad = 0;
    if (n) {                                               // the table of VRCP14 / VRSQRT14 takes sV's place (n is the same in every thread;
        if (tid < 128) const_cast<uint2*>(sTab)[tid] = P.tab14[tid];     // every wave is past its last sV read)
        // Short list: the table is first read by the one-lane hashes, BEHIND the barrier that follows the 16-lane tensors (which touch
        // neither the table nor its place) -- that barrier publishes it (round 6: one workgroup barrier fewer per listed tile, and the
        // table's global round trip overlaps the tensors).  Long list: hash_phase reads the table right away.
        if (n > kListMax) RAISR_BARRIER(tid);
    }
    if (n <= kListMax) {
        // short list (the usual case): 16 lanes per pixel, four pixels per wave and round
        if (n) {
            const int g = lane >> 4, l = lane & 15;
            float wl[11];
#pragma unroll
            for (int i = 0; i < 11; i++) wl[i] = P.gauss_dev[min(l, 10) * 12 + i];
            // two phases: the tensors with 16 lanes per entry (rounds over the waves), parked behind the table in sV's space; then
            // the hash with ONE LANE PER ENTRY -- the hash is ~200 instructions whether 4 or 64 lanes of a wave are active, and
            // in a one-phase loop every wave issues it for its own 4 entries per round (measured, scripts/r03_call31.sh: C1 +9 %,
            // C2 / C3 / C5 +1.5 %)
            float4* sAbd = reinterpret_cast<float4*>(const_cast<uint2*>(sTab) + 128);
#ifdef RAISR_PROBE_L2CERT
            // what a second certification level could save AT MOST: no 16-lane exact tensors (one barrier per listed tile was
            // removed for every build in round 6: the table's own) -- the exact hash runs on the approximate tensors parked at listing time
            // (docs/EXPERIMENTS.md R6)
            sAbd = reinterpret_cast<float4*>(sG);
            if (false)
#endif
            for (unsigned rd = (unsigned)w; 4u * rd < n; rd += 4u) {
                const unsigned e = 4u * rd + (unsigned)g;
                const unsigned ent = sList[min(e, n - 1u)];
                const int prow = (ent >> 6) & 15, pcol = ent & 63;
                float a, b, d;
                exact_tensor16(sG, wl, prow, pcol, l, a, b, d);
                if (l == 0 && e < n) sAbd[e] = float4{a, b, d, 0.0f};
            }
            RAISR_BARRIER(tid);                               // publishes the tensors AND the table
            if (tid < n) {
                const unsigned ent = sList[tid];
                const int prow = (ent >> 6) & 15, pcol = ent & 63;
                const float4 t = sAbd[tid];
                unsigned hA, hB;
                flavour_hash(P, sTab, t.x, t.y, t.z, c0 + pcol, hA, hB);
                if ((ent & 0x8000u) && (sH[prow * TW + pcol] != (uint8_t)hA || (hB != 0xFFu && sH2[prow * TW + pcol] != (uint8_t)hB))) bad++;
                sH[prow * TW + pcol] = (uint8_t)hA;
                sH2[prow * TW + pcol] = (uint8_t)hB;
            }
        }
    } else {
        // long list (synthetic content, self-check mode): the whole tile through the all-exact routine (hash_phase; it
        // rebuilds the gradient tile from the LR window), every wave busy.  The AVX2 flavour takes its out-of-line path.
        unsigned hA[RPW], hB[RPW];
        hash_phase<RPW, false, LW, GT>(P, gw, sL, sG, sTab, nullptr, c0, r0, hA, hB, tid);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int prow = RPW * w + j;
            if (hA[j] != 0xFFu && ((certbits >> j) & 1u) &&
                (sH[prow * TW + lane] != (uint8_t)hA[j] || (hB[j] != 0xFFu && sH2[prow * TW + lane] != (uint8_t)hB[j]))) bad++;
            sH[prow * TW + lane] = (uint8_t)hA[j];
            sH2[prow * TW + lane] = (uint8_t)hB[j];
        }
    }
    if (P.cert_stats && bad) atomicAdd(&sCnt[2], bad);
    if (n || PC) RAISR_BARRIER(tid);                          // (n is the same in every thread)
    RAISR_PHASE(5);                                        // worklist: table staging, exact tensors, exact hashes, three barriers
}

// hash_phase_defer: the hash stage of k_hashfilter_ac<.., DEFER> -- a COMPARISON pipeline (test-hooks / development flavours only:
// RAISR_HIP_DEFER; measured 1-6 % slower than the in-tile worklist of hash_phase_ac, docs/EXPERIMENTS.md R5.1; the product runs
// hash_phase_ac).  As hash_phase_ac up to the certification;
// then every wave is on its own: the pixels it could not certify keep their approximate bucket for the filter stage and go, with
// that bucket, into the wave's region of the frame's fix list (FixAc; k_fix_ac repairs those whose exact bucket differs before
// k_blend reads the HR plane).  A wave with more than kWaveCap of them (synthetic content: 1-px patterns, exact symmetries) runs
// the all-exact code for its four rows instead (hash_rows_exact on the gradient tile, tables from global memory) and lists nothing.
// Against the in-tile worklist of hash_phase_ac: no LDS atomics, no list, no table staging, three workgroup barriers fewer, and
// the 16-lane exact tensors + one-lane hashes of the 1-3 % uncertified pixels leave the issue-bound main kernel for a small
// latency-bound one that overlaps the main kernels of the other frames in flight.  The gradient tile sG stays intact to the end
// of the stage (the symmetric filter stage's zero block lives in the wave's own sV rows instead).
__device__ __forceinline__ unsigned lane_rank(unsigned long long m)      // number of set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <int LW, typename GT>
__device__ __forceinline__ void hash_phase_defer(const PassParams& P, const GaussW& gw, const SepW& S, const GT* sG, typename FVec<4>::type* sV,
                                                 uint8_t* sH, uint8_t* sH2, const FixAc& F, unsigned region, int c0, int r0, unsigned tid = threadIdx.x)
{
    constexpr int TW = 64, RPW = 4;
    const int lane = tid & 63, w = tid >> 6;
    float ta[RPW], tb[RPW], td[RPW];
    RAISR_PHASE_DECL;
    tensor_acN<RPW, GT>(S, sG, sV, ta, tb, td, tid);
    RAISR_PHASE(2);                                        // V pass + barrier + H pass
    const int c = c0 + lane;
    const bool inA = c >= P.a_begin && c < P.a_end, inB = c >= P.b_begin && c < P.b_end;
    const int fl = inB ? 1 : 0;
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    unsigned long long ub[RPW];
    unsigned bk[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        const int prow = RPW * w + j;
        const int r = r0 + prow;
        const bool zone = r < P.H - kMargin && c < P.c_final && (inA || inB);
        unsigned bucket;
        bool cert = approx_hash(ta[j], tb[j], td[j], Q, S, fl, bucket);
        const bool zero = (ta[j] + td[j]) == 0.0f;          // flat window: the reference's tensor is exactly (0, 0, 0) as well
        cert |= zero;
        const unsigned bA = zero ? (unsigned)P.zero_bucket[inA ? 0 : 1] : bucket;
        const unsigned bB = zero ? (unsigned)P.zero_bucket[1] : bucket;
        sH[prow * TW + lane] = zone ? (uint8_t)bA : (uint8_t)0xFFu;
        sH2[prow * TW + lane] = (zone && inA && inB) ? (uint8_t)bB : (uint8_t)0xFFu;
        ub[j] = __ballot(zone && !cert);
        bk[j] = bucket;
    }
    RAISR_PHASE(3);                                        // approximate hash + certification
    unsigned n = 0;
#pragma unroll
    for (int j = 0; j < RPW; j++) n += (unsigned)__builtin_popcountll(ub[j]);
    n = __builtin_amdgcn_readfirstlane(n);
    unsigned count = n;
    if (n > kWaveCap) {                                    // wave-uniform
        unsigned hA[RPW], hB[RPW];
        hash_rows_exact<RPW, false, GT>(P, gw, sG, P.tab14, nullptr, c0, r0, hA, hB, tid);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            sH[(RPW * w + j) * TW + lane] = (uint8_t)hA[j];
            sH2[(RPW * w + j) * TW + lane] = (uint8_t)hB[j];
        }
        count = 0xFFu;
    } else if (n) {
        uint16_t* list = F.entries + (size_t)region * kWaveCap;
        unsigned base = 0;
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            if ((ub[j] >> lane) & 1ull) list[base + lane_rank(ub[j])] = (uint16_t)((unsigned)j | ((unsigned)lane << 2) | (bk[j] << 8));
            base += (unsigned)__builtin_popcountll(ub[j]);
        }
    }
    if (lane == 0) {
        F.counts[region] = (uint8_t)count;
        if (P.cert_stats && n) atomicAdd(&P.cert_stats[0], n);
    }
    RAISR_PHASE(5);                                        // listing / the wave-level all-exact fallback
}

// Test hook: the certified hash stage's decision for arbitrary APPROXIMATE tensor triples (a', b', d'): bucket and whether
// it would be certified, by the very approx_hash the kernels run (flavour 0: AVX-512 table error, 1: AVX2).
__global__ __launch_bounds__(256) void k_debug_approx_hash(const float* __restrict__ abd, unsigned n, PassParams P, SepW S, int fl,
                                                           uint8_t* __restrict__ bucket_out, uint8_t* __restrict__ cert_out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const HashQf Q = {P.qangle, P.qs0, P.qs1, P.qc0, P.qc1};
    const float a = abd[3 * (size_t)i], b = abd[3 * (size_t)i + 1], d = abd[3 * (size_t)i + 2];
    unsigned bucket;
    int si;
    bool ok_str;
    bool cert = approx_hash(a, b, d, Q, S, fl, bucket, &si, &ok_str);
    if ((a + d) == 0.0f) { cert = true; bucket = (unsigned)P.zero_bucket[fl]; }
    else if (P.c1tab && b == 0.0f && (a == 0.0f || d == 0.0f)) {        // exactly one-dimensional window: the class-1 rule, as in hash_phase_ac
        unsigned bA, bB;
        if (class1_buckets(fmaxf(a, d), (unsigned)si, ok_str, S, P, fl == 0, fl == 1, bA, bB)) { cert = true; bucket = bA; }
    }
    bucket_out[i] = (uint8_t)bucket;
    cert_out[i] = cert ? 1 : 0;
}
