/*
 * raisr_oracle_avx512.c -- TEST INFRASTRUCTURE: hand-vectorised AVX-512 twin of raisr_oracle.c's fp32 pass.
 *
 * Purpose: (1) the CPU baseline bench.py prints beside the GPU number ("the build's own AVX-512
 * implementation ... threads = physical cores"), (2) a third independent implementation of the same semantics -- its output
 * must equal raisr_oracle.c's bit for bit (tests/test_oracle_avx512.py, tests/golden/oracle_digests.json).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the shared object this file goes into.
 *
 * Shape (own design; the reference vectorises INSIDE a patch -- one zmm = 16 patch columns of one pixel pair,
 * Library/Raisr_AVX512.cpp:69-131 -- and needs a 16 -> 1 reduction per pixel):
 *   structure tensor   one zmm lane = one PIXEL: 16 adjacent pixels walk the 11 x 11 window together, so every lane runs
 *                      the reference's per-column fma chains (p = gx*w; A = fma(p,gx,A); B = fma(p,gy,B); q = gy*w;
 *                      D = fma(q,gy,D), Raisr_AVX512.cpp:96-121) and the sumitup_ps_512 association (:37-44) on its own
 *                      values: no cross-lane reduction at all.  Gradients are taken once per frame (integer-valued, exact).
 *   hash               16 pixels per call, same operation order as GetHashValue_AVX512_32f_16Elements (:175-258) /
 *                      GetHashValue_AVX256_32f_8Elements (Raisr_AVX256.cpp:393-472).  VRCP14PS / VRSQRT14PS / RCPPS / RSQRTPS
 *                      are NOT executed natively (the GPU box's host is an AMD CPU, whose approximations differ from
 *                      Intel's): their bit-exact integer models (x86_approx.h) are evaluated with vector integer
 *                      arithmetic and gathers.
 *   121-tap filter     one zmm lane = one accumulator lane of the reference (DotProdPatch_AVX512_32f, :134-149): the patch
 *                      is laid out as 8 x 16 floats, 1 mul + 7 fmadd, then the 8/4/2/1 tree.
 *   census blend       flat column loop (compiler-vectorised), as in raisr_oracle.c.
 * Built with -ffp-contract=off -fno-fast-math: every intrinsic below is the one IEEE operation it names.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "x86_approx.h"

#if defined(__FAST_MATH__)
#error "the oracle must be compiled without -ffast-math"
#endif

#define PATCH 11
#define PM 5
#define LM 6
#define TAPS 121

enum { ORA_ASM_AVX2 = 1, ORA_ASM_AVX512 = 2 };
enum { ORA_BLEND_RANDOMNESS = 1, ORA_BLEND_COUNT = 2 };

typedef struct {              /* == ora_pass_t of raisr_oracle.c */
    int bits;
    int lo, hi;
    int pixel_types;
    int asm_type;
    int blending;
    float qangle;
    float qstr[2];
    float qcoh[2];
    const float *bank;
} ora_pass_t;

/* from raisr_oracle.c (same shared object) */
void ora_gaussian_weights(int bits, float w[PATCH][PATCH]);
void ora_resize_bilinear(const uint16_t *src, int sw, int sh, int sstride, uint16_t *dst, int dw, int dh, int dstride, int tie);
void ora_pass(const uint16_t *lr, int W, int H, const ora_pass_t *P, uint16_t *out, int32_t *hash_dump, float *hr_dump);

/* ---- approximation-instruction models, 16 lanes at a time ------------------------------------------------------------- */
static uint32_t T14_C0[128], T14_C1[128];            /* [0,64): VRCP14 rows, [64,128): VRSQRT14 rows [parity][top 5 mantissa bits] */
static uint32_t TLEG[4096];                          /* [0,2048): RCPPS LUT, [2048,4096): RSQRTPS LUT [parity][m >> 13] */
static int tables_ready;

static void init_tables(void)
{
    if (tables_ready) return;
    for (int i = 0; i < 64; i++) {
        T14_C0[i] = X86_RCP14_C0[i]; T14_C1[i] = X86_RCP14_C1[i];
        T14_C0[64 + i] = X86_RSQRT14_C0[i]; T14_C1[64 + i] = X86_RSQRT14_C1[i];
    }
    for (int i = 0; i < 2048; i++) { TLEG[i] = X86_RCP_LUT[i]; TLEG[2048 + i] = X86_RSQRT_LUT[i]; }
    tables_ready = 1;
}

/* VRCP14PS(VRSQRT14PS(v)) for 16 lanes.  Straight-line for +normal, +-0 and negative inputs; lanes holding a NaN, +inf or a
 * positive denormal (never produced by 8/10-bit content) are redone with the scalar models. */
static inline __m512 sqrt14_ps(__m512 v)
{
    const __m512i x = _mm512_castps_si512(v);
    /* VRSQRT14: row = [p = ~E & 1][m >> 18] = bits 23..18 of x with bit 23 inverted; t = (m >> 8) & 1023 */
    const __m512i row = _mm512_add_epi32(_mm512_xor_si512(_mm512_and_si512(_mm512_srli_epi32(x, 18), _mm512_set1_epi32(63)), _mm512_set1_epi32(32)),
                                         _mm512_set1_epi32(64));
    const __m512i t = _mm512_and_si512(_mm512_srli_epi32(x, 8), _mm512_set1_epi32(1023));
    __m512i code = _mm512_srli_epi32(_mm512_sub_epi32(_mm512_i32gather_epi32(row, T14_C0, 4),
                                                      _mm512_mullo_epi32(_mm512_i32gather_epi32(row, T14_C1, 4), t)), 9);
    const __mmask16 pow4 = _mm512_cmpeq_epi32_mask(_mm512_and_si512(x, _mm512_set1_epi32(0x00ffffff)), _mm512_set1_epi32(0x00800000));
    code = _mm512_mask_mov_epi32(code, pow4, _mm512_setzero_si512());       /* p == 0 && m == 0: exact power of four */
    /* VRCP14 of y = 2^(-half-1) (1 + code / 65536): its row is code >> 10, its t is code & 1023 */
    const __m512i row2 = _mm512_srli_epi32(code, 10);
    const __m512i code2 = _mm512_srli_epi32(_mm512_sub_epi32(_mm512_i32gather_epi32(row2, T14_C0, 4),
                                                             _mm512_mullo_epi32(_mm512_i32gather_epi32(row2, T14_C1, 4),
                                                                                _mm512_and_si512(code, _mm512_set1_epi32(1023)))), 9);
    const __m512i ez = _mm512_and_si512(_mm512_srli_epi32(_mm512_add_epi32(x, _mm512_set1_epi32(0x3f800000)), 1), _mm512_set1_epi32(0x7f800000));
    const __m512i zp = _mm512_mask_mov_epi32(_mm512_add_epi32(ez, _mm512_set1_epi32(0x00800000)), pow4, ez);
    const __mmask16 code0 = _mm512_cmpeq_epi32_mask(code, _mm512_setzero_si512());
    __m512i z = _mm512_mask_mov_epi32(_mm512_or_si512(ez, _mm512_slli_epi32(code2, 7)), code0, zp);
    /* classes (on the bit pattern): +normal -> z; +-0 -> x; negative and not NaN (normal, denormal, -inf) -> QNaN indefinite;
     * NaN / +inf / +denormal -> scalar models */
    const __m512i mag = _mm512_and_si512(x, _mm512_set1_epi32(0x7fffffff));
    const __mmask16 sign = _mm512_cmplt_epi32_mask(x, _mm512_setzero_si512());
    const __mmask16 zero = _mm512_cmpeq_epi32_mask(mag, _mm512_setzero_si512());
    const __mmask16 nan = _mm512_cmpgt_epi32_mask(mag, _mm512_set1_epi32(0x7f800000));
    const __mmask16 negative = sign & (__mmask16)~zero & (__mmask16)~nan;
    const __mmask16 pinf = _mm512_cmpeq_epi32_mask(x, _mm512_set1_epi32(0x7f800000));
    const __mmask16 pden = (__mmask16)~sign & (__mmask16)~zero & _mm512_cmplt_epi32_mask(mag, _mm512_set1_epi32(0x00800000));
    const __mmask16 rare = nan | pinf | pden;
    z = _mm512_mask_mov_epi32(z, negative, _mm512_set1_epi32((int)0xffc00000u));
    z = _mm512_mask_mov_epi32(z, zero, x);
    __m512 out = _mm512_castsi512_ps(z);
    if (rare) {
        float tmp[16] __attribute__((aligned(64))), src[16] __attribute__((aligned(64)));
        _mm512_store_ps(tmp, out); _mm512_store_ps(src, v);
        for (int l = 0; l < 16; l++) if ((rare >> l) & 1) tmp[l] = x86_rcp14(x86_rsqrt14(src[l]));
        out = _mm512_load_ps(tmp);
    }
    return out;
}

/* RCPPS(RSQRTPS(v)) for 16 lanes, every class in line (x86_rcp / x86_rsqrt of x86_approx.h composed) */
static inline __m512 sqrt_legacy_ps(__m512 v)
{
    const __m512i x = _mm512_castps_si512(v);
    const __m512i qi = _mm512_add_epi32(_mm512_xor_si512(_mm512_and_si512(_mm512_srli_epi32(x, 13), _mm512_set1_epi32(2047)), _mm512_set1_epi32(1024)),
                                        _mm512_set1_epi32(2048));
    const __m512i q = _mm512_i32gather_epi32(qi, TLEG, 4);                   /* RSQRTPS mantissa code (12 bits) */
    const __m512i r = _mm512_i32gather_epi32(_mm512_srli_epi32(q, 1), TLEG, 4);
    const __m512i ez = _mm512_and_si512(_mm512_srli_epi32(_mm512_add_epi32(x, _mm512_set1_epi32(0x3f800000)), 1), _mm512_set1_epi32(0x7f800000));
    __m512i z = _mm512_or_si512(ez, _mm512_slli_epi32(r, 11));
    const __m512i mag = _mm512_and_si512(x, _mm512_set1_epi32(0x7fffffff));
    const __mmask16 sign = _mm512_cmplt_epi32_mask(x, _mm512_setzero_si512());
    const __mmask16 nan = _mm512_cmpgt_epi32_mask(mag, _mm512_set1_epi32(0x7f800000));
    const __mmask16 tiny = _mm512_cmplt_epi32_mask(mag, _mm512_set1_epi32(0x00800000));              /* +-0, +-denormal (DAZ) */
    const __mmask16 neg = sign & (__mmask16)~tiny & (__mmask16)~nan;                                  /* -inf, negative normal */
    const __mmask16 pinf = _mm512_cmpeq_epi32_mask(x, _mm512_set1_epi32(0x7f800000));
    z = _mm512_mask_mov_epi32(z, pinf, x);
    z = _mm512_mask_mov_epi32(z, nan, _mm512_or_si512(x, _mm512_set1_epi32(0x00400000)));
    z = _mm512_mask_mov_epi32(z, neg, _mm512_set1_epi32((int)0xffc00000u));
    z = _mm512_mask_mov_epi32(z, tiny, _mm512_and_si512(x, _mm512_set1_epi32((int)0x80000000u)));
    return _mm512_castsi512_ps(z);
}

/* 16 hash buckets from 16 (a, b, d); legacy = the AVX2 flavour (Raisr_AVX256.cpp:393-472) */
static inline __m512i hash16(__m512 a, __m512 b, __m512 d, const ora_pass_t *P, int legacy)
{
    const __m512 zero = _mm512_setzero_ps();
    const __m512 T = _mm512_add_ps(a, d);
    const __m512 Dt = _mm512_sub_ps(_mm512_mul_ps(a, d), _mm512_mul_ps(b, b));
    const __m512 rad = _mm512_sub_ps(_mm512_mul_ps(_mm512_mul_ps(T, T), _mm512_set1_ps(0.25f)), Dt);     /* x / 4 == x * 0.25 exactly */
    const __m512 s = legacy ? sqrt_legacy_ps(rad) : sqrt14_ps(rad);
    const __m512 hT = _mm512_mul_ps(T, _mm512_set1_ps(0.5f));
    const __m512 L1 = _mm512_add_ps(hT, s), L2 = _mm512_sub_ps(hT, s);
    const __mmask16 bnz = _mm512_cmp_ps_mask(b, zero, _CMP_NEQ_OQ);
    const __m512 xx = _mm512_mask_mov_ps(_mm512_set1_ps(1.0f), bnz, _mm512_sub_ps(L1, d));
    /* atan2 approximation (Raisr_AVX512.cpp:151-173): r = x < 0 ? (x + |y|') / (|y|' - x) : (x - |y|') / (x + |y|') */
    const __m512 ay = _mm512_add_ps(_mm512_abs_ps(b), _mm512_set1_ps(1e-10f));
    const __mmask16 neg = _mm512_cmp_ps_mask(xx, zero, _CMP_LT_OQ);
    const __m512 xpa = _mm512_add_ps(xx, ay);
    const __m512 num = _mm512_mask_mov_ps(_mm512_sub_ps(xx, ay), neg, xpa);
    const __m512 den = _mm512_mask_mov_ps(xpa, neg, _mm512_sub_ps(ay, xx));
    const __m512 rr = _mm512_div_ps(num, den);
    __m512 ang = _mm512_mask_mov_ps(_mm512_set1_ps((float)(M_PI / 4.0)), neg, _mm512_set1_ps((float)(3.0 * M_PI / 4.0)));
    ang = _mm512_fmadd_ps(_mm512_fmadd_ps(_mm512_mul_ps(_mm512_set1_ps(0.1963f), rr), rr, _mm512_set1_ps(-0.9817f)), rr, ang);
    const __m512 nang = _mm512_mul_ps(_mm512_set1_ps(-1.0f), ang);
    ang = _mm512_mask_mov_ps(ang, _mm512_cmp_ps_mask(b, zero, _CMP_LT_OQ), nang);
    ang = _mm512_add_ps(ang, _mm512_mask_mov_ps(zero, _mm512_cmp_ps_mask(ang, zero, _CMP_LT_OQ), _mm512_set1_ps(3.141592653f)));
    const __m512 sL1 = legacy ? sqrt_legacy_ps(L1) : sqrt14_ps(L1);
    const __m512 sL2 = legacy ? sqrt_legacy_ps(L2) : sqrt14_ps(L2);
    const __m512 coh = _mm512_div_ps(_mm512_sub_ps(sL1, sL2), _mm512_add_ps(_mm512_add_ps(sL1, sL2), _mm512_set1_ps(1e-17f)));
    const __m512 str = L1;
    __m512i ai = _mm512_cvtps_epi32(_mm512_floor_ps(_mm512_mul_ps(ang, _mm512_set1_ps(P->qangle))));     /* NaN / range -> INT_MIN */
    ai = _mm512_min_epi32(_mm512_max_epi32(ai, _mm512_setzero_si512()), _mm512_set1_epi32(23));
    const __m512i one = _mm512_set1_epi32(1);
    __m512i si, ci;
    if (!legacy) {
        si = _mm512_add_epi32(_mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(_mm512_set1_ps(P->qstr[0]), str, _CMP_LE_OQ), one),
                              _mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(_mm512_set1_ps(P->qstr[1]), str, _CMP_LE_OQ), one));
        ci = _mm512_add_epi32(_mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(_mm512_set1_ps(P->qcoh[0]), coh, _CMP_LE_OQ), one),
                              _mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(_mm512_set1_ps(P->qcoh[1]), coh, _CMP_LE_OQ), one));
    } else {
        si = _mm512_sub_epi32(_mm512_set1_epi32(2),
                              _mm512_add_epi32(_mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(str, _mm512_set1_ps(P->qstr[0]), _CMP_LE_OQ), one),
                                               _mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(str, _mm512_set1_ps(P->qstr[1]), _CMP_LE_OQ), one)));
        ci = _mm512_sub_epi32(_mm512_set1_epi32(2),
                              _mm512_add_epi32(_mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(coh, _mm512_set1_ps(P->qcoh[0]), _CMP_LE_OQ), one),
                                               _mm512_maskz_mov_epi32(_mm512_cmp_ps_mask(coh, _mm512_set1_ps(P->qcoh[1]), _CMP_LE_OQ), one)));
    }
    return _mm512_add_epi32(_mm512_add_epi32(_mm512_mullo_epi32(ai, _mm512_set1_epi32(9)), _mm512_mullo_epi32(si, _mm512_set1_epi32(3))), ci);
}

/* ---- structure tensor of 16 adjacent pixels (row r, columns x .. x+15), lane = pixel -------------------------------------- */
static inline void tensor16(const float *GX, const float *GY, size_t stride, int r, int x, const float w[PATCH][PATCH],
                            __m512 *pa, __m512 *pb, __m512 *pd)
{
    __m512 SA[PATCH], SB[PATCH], SD[PATCH];
    for (int k = 0; k < PATCH; k++) {
        __m512 A = _mm512_setzero_ps(), B = _mm512_setzero_ps(), D = _mm512_setzero_ps();
        const float *gxp = GX + (size_t)(r - PM) * stride + (x - PM + k);
        const float *gyp = GY + (size_t)(r - PM) * stride + (x - PM + k);
        for (int i = 0; i < PATCH; i++) {
            const __m512 gx = _mm512_loadu_ps(gxp + (size_t)i * stride), gy = _mm512_loadu_ps(gyp + (size_t)i * stride);
            const __m512 wv = _mm512_set1_ps(w[i][k]);
            const __m512 p = _mm512_mul_ps(gx, wv);
            A = _mm512_fmadd_ps(p, gx, A);
            B = _mm512_fmadd_ps(p, gy, B);
            const __m512 q = _mm512_mul_ps(gy, wv);
            D = _mm512_fmadd_ps(q, gy, D);
        }
        SA[k] = A; SB[k] = B; SD[k] = D;
    }
    /* sumitup_ps_512 on the even / odd lane placements: (Gb + Gc) + (Ga + Gd) up to the operand order of single additions */
#define FOLD(S) _mm512_add_ps(_mm512_add_ps(_mm512_add_ps(S[7], S[3]), _mm512_add_ps(_mm512_add_ps(S[1], S[9]), S[5])), \
                              _mm512_add_ps(_mm512_add_ps(_mm512_add_ps(S[0], S[8]), S[4]), _mm512_add_ps(_mm512_add_ps(S[2], S[10]), S[6])))
    *pa = FOLD(SA); *pb = FOLD(SB); *pd = FOLD(SD);
#undef FOLD
}

/* ---- DotProdPatch: patch rows r-5..r+5, columns c-5..c+5 of L against one padded bank row -------------------------------- */
/* The reference stores the 11 patch rows into a 128-float buffer and reloads it as 8 vectors (Raisr_AVX512.cpp:116-117); here the
 * 8 vectors of taps 16 ch .. 16 ch + 15 are permuted straight out of the 11 row registers (a reload right after 44-byte stores
 * cannot be store-forwarded).  Tap k lives in row k / 11, column k % 11; a chunk spans two or three rows. */
#define DOT_GROUP 4                   /* pixels served by one set of 11 row loads: 16 floats per row = columns c-5 .. c+10 */
static __m512i PIDX[DOT_GROUP][8][2]; /* per pixel of the group and chunk: indices into the first two rows (permutex2var) / the third row */
static __mmask16 PMASK3[8];           /* lanes taken from the third row */
static int PROW[8][3];
static int perm_ready;

static void init_perm(void)
{
    if (perm_ready) return;
    for (int sh = 0; sh < DOT_GROUP; sh++)
        for (int ch = 0; ch < 8; ch++) {
            int32_t i1[16] = {0}, i2[16] = {0};
            int ra = (16 * ch) / PATCH, rb = ra + 1, rc = ra + 2;
            __mmask16 m3 = 0;
            for (int l = 0; l < 16; l++) {
                const int k = 16 * ch + l;
                if (k >= TAPS) { i1[l] = 0; continue; }             /* padding lanes: any finite sample, the coefficient is +0 */
                const int row = k / PATCH, col = k % PATCH + sh;
                if (row == ra) i1[l] = col;
                else if (row == rb) i1[l] = 16 + col;
                else { i2[l] = col; m3 |= (__mmask16)(1u << l); }
            }
            if (rb > PATCH - 1) rb = PATCH - 1;
            if (rc > PATCH - 1) rc = PATCH - 1;
            PROW[ch][0] = ra; PROW[ch][1] = rb; PROW[ch][2] = rc;
            PIDX[sh][ch][0] = _mm512_loadu_si512(i1); PIDX[sh][ch][1] = _mm512_loadu_si512(i2);
            PMASK3[ch] = m3;
        }
    perm_ready = 1;
}

/* the 11 window rows of pixels c .. c + DOT_GROUP - 1 of row r */
static inline void dot_rows(const float *L, size_t stride, int r, int c, __m512 R[PATCH])
{
    const float *p = L + (size_t)(r - PM) * stride + (c - PM);
    for (int i = 0; i < PATCH; i++) R[i] = _mm512_loadu_ps(p + (size_t)i * stride);
}

/* pixel c + sh of the group whose rows are in R, against one padded bank row */
static inline float dot121(const __m512 R[PATCH], int sh, const float *f)
{
    __m512 acc = _mm512_setzero_ps();
    for (int ch = 0; ch < 8; ch++) {
        __m512 v = _mm512_permutex2var_ps(R[PROW[ch][0]], PIDX[sh][ch][0], R[PROW[ch][1]]);
        if (PMASK3[ch]) v = _mm512_mask_permutexvar_ps(v, PMASK3[ch], PIDX[sh][ch][1], R[PROW[ch][2]]);
        const __m512 fv = _mm512_load_ps(f + 16 * ch);
        acc = ch == 0 ? _mm512_mul_ps(v, fv) : _mm512_fmadd_ps(v, fv, acc);
    }
    const __m256 t = _mm256_add_ps(_mm512_castps512_ps256(acc), _mm512_extractf32x8_ps(acc, 1));       /* a[i] + a[i+8] */
    const __m128 u = _mm_add_ps(_mm256_castps256_ps128(t), _mm256_extractf128_ps(t, 1));                /* t[i] + t[i+4] */
    const __m128 s = _mm_add_ps(u, _mm_movehl_ps(u, u));                                                /* (u0+u2, u1+u3) */
    return _mm_cvtss_f32(_mm_add_ss(s, _mm_movehdup_ps(s)));                                            /* (u0+u2) + (u1+u3) */
}

/* One RAISR pass, CountOfBitsChanged blending (Randomness goes to the scalar oracle). */
void ora512_pass(const uint16_t *lr, int W, int H, const ora_pass_t *P, uint16_t *out)
{
    if (P->blending == ORA_BLEND_RANDOMNESS || W < 2 * LM + 1 || H < 2 * LM + 1) { ora_pass(lr, W, H, P, out, NULL, NULL); return; }
    init_tables();
    init_perm();
    float wg[PATCH][PATCH];
    ora_gaussian_weights(P->bits, wg);
    const size_t stride = (size_t)W, n = stride * H, slack = 64;
    /* work planes kept between calls (a fresh 4K plane costs more in page faults than the pass spends on it); calls are
     * serialised by the callers (tests, bench.py), this is not a re-entrant library */
    static float *planes[4];
    static size_t plane_cap;
    if (plane_cap < n + slack) {
        for (int i = 0; i < 4; i++) { free(planes[i]); planes[i] = (float *)aligned_alloc(64, ((n + slack) * sizeof(float) + 63) & ~(size_t)63); }
        plane_cap = n + slack;
    }
    float *L = planes[0], *HR = planes[1], *GX = planes[2], *GY = planes[3];
    const int rows = 216 * P->pixel_types;
    float *bank = (float *)aligned_alloc(64, (size_t)rows * 128 * sizeof(float));
    for (int i = 0; i < rows; i++) {
        memcpy(bank + (size_t)i * 128, P->bank + (size_t)i * TAPS, TAPS * sizeof(float));
        memset(bank + (size_t)i * 128 + TAPS, 0, (128 - TAPS) * sizeof(float));
    }
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const uint16_t *s = lr + (size_t)y * W;
        float *l = L + (size_t)y * stride, *h = HR + (size_t)y * stride;
        for (int x = 0; x < W; x++) { l[x] = (float)s[x]; h[x] = l[x]; }
    }
    for (size_t i = n; i < n + slack; i++) L[i] = GX[i] = GY[i] = 0.0f;
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        float *gx = GX + (size_t)y * stride, *gy = GY + (size_t)y * stride;
        if (y == 0 || y == H - 1) { memset(gx, 0, W * sizeof(float)); memset(gy, 0, W * sizeof(float)); continue; }
        const float *up = L + (size_t)(y - 1) * stride, *mid = L + (size_t)y * stride, *dn = L + (size_t)(y + 1) * stride;
        gx[0] = gy[0] = gx[W - 1] = gy[W - 1] = 0.0f;
        for (int x = 1; x < W - 1; x++) { gx[x] = dn[x] - up[x]; gy[x] = mid[x + 1] - mid[x - 1]; }     /* GetGx / GetGy */
    }
    const float lo = (float)P->lo, hi = (float)P->hi;
    #pragma omp parallel
    {
        float *ta = (float *)aligned_alloc(64, 3 * (((size_t)W + 31) & ~(size_t)15) * sizeof(float));
        float *tb = ta + (((size_t)W + 31) & ~(size_t)15), *td = tb + (((size_t)W + 31) & ~(size_t)15);
        #pragma omp for schedule(dynamic, 2)
        for (int r = LM; r < H - LM; r++) {
            for (int x = LM; x < W - LM; x += 16) {
                __m512 a, b, d;
                tensor16(GX, GY, stride, r, x, wg, &a, &b, &d);
                _mm512_storeu_ps(ta + x, a); _mm512_storeu_ps(tb + x, b); _mm512_storeu_ps(td + x, d);
            }
            const int unroll = P->asm_type == ORA_ASM_AVX512 ? 16 : 8;        /* column driver of Raisr.cpp:1058-1250 */
            int loopItr = unroll, c = LM;
            while (c + loopItr <= W - LM) {
                const int legacy = loopItr == 8;
                int32_t hb[16] __attribute__((aligned(64)));
                _mm512_store_si512((__m512i *)hb, hash16(_mm512_loadu_ps(ta + c), _mm512_loadu_ps(tb + c), _mm512_loadu_ps(td + c), P, legacy));
                for (int g = 0; g < loopItr; g += DOT_GROUP) {       /* loopItr is 16 or 8 */
                    __m512 R[PATCH];
                    dot_rows(L, stride, r, c + g, R);
                    for (int sh = 0; sh < DOT_GROUP; sh++) {
                        const int pix = g + sh, cc = c + pix;
                        int t = 0;
                        if (P->pixel_types == 4) t = ((r - PM) % 2) * 2 + ((cc - PM) % 2);
                        const float v = dot121(R, sh, bank + ((size_t)hb[pix] * P->pixel_types + t) * 128);
                        if (v > lo && v < hi) HR[(size_t)r * stride + cc] = v;
                    }
                }
                if (loopItr > 8 && c + 2 * unroll > W - LM) loopItr = 8;
                c += loopItr;
            }
        }
        free(ta);
    }
    /* borders + census blend: as raisr_oracle.c */
    for (int c = 0; c < W; c++) { out[c] = lr[c]; out[(size_t)(H - 1) * W + c] = lr[(size_t)(H - 1) * W + c]; }
    for (int r = 0; r < H; r++) { out[(size_t)r * W] = lr[(size_t)r * W]; out[(size_t)r * W + W - 1] = lr[(size_t)r * W + W - 1]; }
    #pragma omp parallel for schedule(static)
    for (int r = 1; r < H - 1; r++) {
        const float *l0 = L + (size_t)(r - 1) * stride, *l1 = L + (size_t)r * stride, *l2 = L + (size_t)(r + 1) * stride;
        const float *h0 = HR + (size_t)(r - 1) * stride, *h1 = HR + (size_t)r * stride, *h2 = HR + (size_t)(r + 1) * stride;
        uint16_t *o = out + (size_t)r * W;
        const int ilo = P->lo, ihi = P->hi;
        for (int c = 1; c < W - 1; c++) {
            const float lc = l1[c], hc = h1[c];
            int hd = 0;
#define ORA_CT(ln, hn) hd += ((ln) < lc) != ((hn) < hc)
            ORA_CT(l0[c - 1], h0[c - 1]); ORA_CT(l0[c], h0[c]); ORA_CT(l0[c + 1], h0[c + 1]);
            ORA_CT(l1[c - 1], h1[c - 1]);                       ORA_CT(l1[c + 1], h1[c + 1]);
            ORA_CT(l2[c - 1], h2[c - 1]); ORA_CT(l2[c], h2[c]); ORA_CT(l2[c + 1], h2[c + 1]);
#undef ORA_CT
            const float weight = (float)hd / 8.0f;
            const float w2 = 1.0f - weight;
            float val = (weight * lc) + (w2 * hc);
            val = val + 0.5f;
            const float fl = floorf(val);
            int32_t iv = (fl >= -2147483648.0f && fl < 2147483648.0f) ? (int32_t)fl : INT32_MIN;
            if (iv > ihi) iv = ihi;
            if (iv < ilo) iv = ilo;
            o[c] = (uint16_t)iv;
        }
    }
    free(bank);
}

/* whole Y-plane job: ora_process_y with ora512_pass */
void ora512_process_y(const uint16_t *in, int inW, int inH, uint16_t *out, int outW, int outH,
                      int passes, int mode, const ora_pass_t *P1, const ora_pass_t *P2, int tie)
{
    if (passes == 1) {
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        ora_resize_bilinear(in, inW, inH, inW, lr, outW, outH, outW, tie);
        ora512_pass(lr, outW, outH, P1, out);
        free(lr);
        return;
    }
    if (mode == 2) {
        uint16_t *mid = (uint16_t *)calloc((size_t)inW * inH, sizeof(uint16_t));
        ora512_pass(in, inW, inH, P1, mid);
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        ora_resize_bilinear(mid, inW, inH, inW, lr, outW, outH, outW, tie);
        ora512_pass(lr, outW, outH, P2, out);
        free(mid); free(lr);
    } else {
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        uint16_t *mid = (uint16_t *)calloc((size_t)outW * outH, sizeof(uint16_t));
        ora_resize_bilinear(in, inW, inH, inW, lr, outW, outH, outW, tie);
        ora512_pass(lr, outW, outH, P1, mid);
        ora512_pass(mid, outW, outH, P2, out);
        free(lr); free(mid);
    }
}
