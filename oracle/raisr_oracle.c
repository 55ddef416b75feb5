/*
 * raisr_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's Enhanced-RAISR
 * Y-plane hot path (whole-frame semantics == the reference run with threadcount=1).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (libraisr_hip) never links, includes or calls anything here.
 *
 * PARITY UNPINNED: the reference has no golden vectors and cannot be built in
 * this image -- every translation unit includes <ipp.h> (reference Library/Raisr_globals.h:9,
 * Library/Raisr.cpp:16) and the cheap-upscale step is Intel IPP's closed ippiResizeLinear
 * (Library/Raisr.cpp:947-958).  What IS pinned: the x86 approximation instructions the hash
 * depends on are modelled bit-exactly and verified exhaustively against a real Intel AVX-512
 * core (x86_approx.h); everything else is a line-cited scalar restatement of the reference's
 * "strict source" semantics (no -ffast-math contraction/reassociation), written so each
 * floating-point operation below corresponds to exactly one intrinsic of the cited lines.
 *
 * Semantics restated (fp32 paths; the fp16 path lives in raisr_oracle_fp16.c):
 *   - cheap upscale: BUILD-DEFINED stand-in for ippiResizeLinear_{8u,16u}_C1R with
 *     ippBorderRepl (Library/Raisr.cpp:947-958,1373-1388): centre-aligned bilinear in exact
 *     integer arithmetic, selectable tie rule.
 *   - per-pixel structure tensor: computeGTWG_Segment_AVX512_32f, Library/Raisr_AVX512.cpp:69-131
 *     (AVX2 twin Library/Raisr_AVX256.cpp:249-337 has the same association).
 *   - hash: GetHashValue_AVX512_32f_16Elements Library/Raisr_AVX512.cpp:175-258 and
 *     GetHashValue_AVX256_32f_8Elements Library/Raisr_AVX256.cpp:393-472, with
 *     atan2Approximation (USE_ATAN2_APPROX build) :151-173 / :368-391.
 *   - filter: DotProdPatch_AVX512_32f Library/Raisr_AVX512.cpp:134-149 + accept test
 *     Library/Raisr.cpp:1196-1200.
 *   - column-chunk driver incl. the 16->8 tail re-computation: Library/Raisr.cpp:1058-1250.
 *   - borders: Library/Raisr.cpp:999-1036,1252-1265.
 *   - blend: CTCountOfBitsChangedSegment_AVX256_32f Library/Raisr_AVX256.cpp:68-166;
 *     Randomness blend Library/Raisr.cpp:1203-1242 with CTRandomness_AVX512_32f
 *     Library/Raisr_AVX512.cpp:19-35.
 *   - two-pass orchestration: Library/Raisr.cpp:896-975, RNLSetRes :1703-1723.
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off -fno-fast-math are REQUIRED).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "x86_approx.h"

#if defined(__FAST_MATH__)
#error "the oracle must be compiled without -ffast-math"
#endif

#define PATCH 11
#define PM 5      /* gPatchMargin, Library/Raisr.cpp:1573 */
#define LM 6      /* gLoopMargin,  Library/Raisr.cpp:1574 */
#define TAPS 121

enum { ORA_ASM_AVX2 = 1, ORA_ASM_AVX512 = 2 };            /* ASMType values, RaisrDefaults.h:37-44 */
enum { ORA_BLEND_RANDOMNESS = 1, ORA_BLEND_COUNT = 2 };   /* BlendingMode, RaisrDefaults.h:31-35 */
enum { ORA_TIE_HALF_UP = 0, ORA_TIE_HALF_EVEN = 1 };

typedef struct {
    int bits;            /* 8, 10 or 16 */
    int lo, hi;          /* gMin/gMax, Library/Raisr.cpp:1451-1468 */
    int pixel_types;     /* 4 when ratio == 2 (gUsePixelType), else 1; Library/Raisr.cpp:1477-1480 */
    int asm_type;        /* ORA_ASM_* */
    int blending;        /* ORA_BLEND_* */
    float qangle;        /* gQAngle = 24 / PI, Library/Raisr.cpp:1553 */
    float qstr[2];       /* Qfactor_strbin thresholds */
    float qcoh[2];       /* Qfactor_cohbin thresholds */
    const float *bank;   /* [216][pixel_types][121], file order (Library/Raisr.cpp:336-340) */
} ora_pass_t;

/* ------------------------------------------------------------------------------------------
 * A.1 cheap upscale (build-defined).  Per axis: n = (2d+1)*S - D, den = 2D, i0 = floor(n/den),
 * f = n - i0*den; taps clamp(i0), clamp(i0+1) with weights (den-f, f).
 * ------------------------------------------------------------------------------------------ */
static inline void axis_tap(int d, int S, int D, int *i0, int *i1, int64_t *f)
{
    int64_t n = (int64_t)(2 * d + 1) * S - D, den = 2 * (int64_t)D;
    int64_t q = n >= 0 ? n / den : -((-n + den - 1) / den);
    *f = n - q * den;
    int a = (int)q, b = (int)q + 1;
    if (a < 0) a = 0; if (a > S - 1) a = S - 1;
    if (b < 0) b = 0; if (b > S - 1) b = S - 1;
    *i0 = a; *i1 = b;
}

static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

void ora_resize_bilinear_generic(const uint16_t *src, int sw, int sh, int sstride,
                                 uint16_t *dst, int dw, int dh, int dstride, int tie);

/* Same arithmetic with the ratios reduced by their gcd (n / den is unchanged by a common factor) whenever every
 * intermediate then fits 32 bits: the two 64-bit divisions per pixel of the generic form become one multiply-shift
 * (floor(t / d) = (t * ceil(2^40 / d)) >> 40, exact for t < 2^24 and d < 2^16).  tests/test_oracle_facts.py compares the
 * two forms on random geometries.  (Speed of the CPU baseline only.) */
void ora_resize_bilinear(const uint16_t *src, int sw, int sh, int sstride,
                         uint16_t *dst, int dw, int dh, int dstride, int tie)
{
    const int64_t gx = gcd64(sw, dw), gy = gcd64(sh, dh);
    const int64_t Sx = sw / gx, Dx = dw / gx, Sy = sh / gy, Dy = dh / gy;
    const int64_t denx = 2 * Dx, deny = 2 * Dy, den = denx * deny;
    if (2 * den * 65535 + den >= (1 << 24) || 2 * den >= 65536) {
        ora_resize_bilinear_generic(src, sw, sh, sstride, dst, dw, dh, dstride, tie);
        return;
    }
    int *x0 = (int *)malloc(sizeof(int) * dw), *x1 = (int *)malloc(sizeof(int) * dw);
    uint32_t *fx = (uint32_t *)malloc(sizeof(uint32_t) * dw);
    for (int x = 0; x < dw; x++) {
        int64_t n = (int64_t)(2 * x + 1) * Sx - Dx;
        int64_t q = n >= 0 ? n / denx : -((-n + denx - 1) / denx);
        fx[x] = (uint32_t)(n - q * denx);
        int a = (int)q, b = (int)q + 1;
        if (a < 0) a = 0; if (a > sw - 1) a = sw - 1;
        if (b < 0) b = 0; if (b > sw - 1) b = sw - 1;
        x0[x] = a; x1[x] = b;
    }
    const uint32_t d2 = (uint32_t)(2 * den), udenx = (uint32_t)denx, udeny = (uint32_t)deny, uden = (uint32_t)den;
    const uint64_t M = ((1ull << 40) + d2 - 1) / d2;
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        int64_t n = (int64_t)(2 * y + 1) * Sy - Dy;
        int64_t qy = n >= 0 ? n / deny : -((-n + deny - 1) / deny);
        const uint32_t fy = (uint32_t)(n - qy * deny);
        int y0 = (int)qy, y1 = (int)qy + 1;
        if (y0 < 0) y0 = 0; if (y0 > sh - 1) y0 = sh - 1;
        if (y1 < 0) y1 = 0; if (y1 > sh - 1) y1 = sh - 1;
        const uint16_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
        uint16_t *o = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            const uint32_t top = (udenx - fx[x]) * r0[x0[x]] + fx[x] * r0[x1[x]];
            const uint32_t bot = (udenx - fx[x]) * r1[x0[x]] + fx[x] * r1[x1[x]];
            const uint32_t num = (udeny - fy) * top + fy * bot;
            const uint32_t t = 2 * num + uden;
            uint32_t q = (uint32_t)(((uint64_t)t * M) >> 40);              /* round half up */
            if (tie == ORA_TIE_HALF_EVEN && t - q * d2 == 0 && (q & 1)) q--;
            o[x] = (uint16_t)q;
        }
    }
    free(x0); free(x1); free(fx);
}

void ora_resize_bilinear_generic(const uint16_t *src, int sw, int sh, int sstride,
                                 uint16_t *dst, int dw, int dh, int dstride, int tie)
{
    const int64_t denx = 2 * (int64_t)dw, deny = 2 * (int64_t)dh;
    int *x0 = (int *)malloc(sizeof(int) * dw), *x1 = (int *)malloc(sizeof(int) * dw);
    int64_t *fx = (int64_t *)malloc(sizeof(int64_t) * dw);
    for (int x = 0; x < dw; x++) axis_tap(x, sw, dw, &x0[x], &x1[x], &fx[x]);
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        int y0, y1; int64_t fy;
        axis_tap(y, sh, dh, &y0, &y1, &fy);
        const uint16_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
        for (int x = 0; x < dw; x++) {
            int64_t top = (denx - fx[x]) * r0[x0[x]] + fx[x] * r0[x1[x]];
            int64_t bot = (denx - fx[x]) * r1[x0[x]] + fx[x] * r1[x1[x]];
            int64_t num = (deny - fy) * top + fy * bot;
            int64_t den = denx * deny;
            int64_t q = (2 * num + den) / (2 * den);                 /* round half up */
            if (tie == ORA_TIE_HALF_EVEN && (2 * num + den) % (2 * den) == 0 && (q & 1)) q--;
            dst[(size_t)y * dstride + x] = (uint16_t)q;
        }
    }
    free(x0); free(x1); free(fx);
}

/* ------------------------------------------------------------------------------------------
 * A.2 Gaussian weights: gGaussian2D{8,10,16}bit, Library/Raisr_globals.h:208-264.  The literal
 * table is symmetric in both axes; Q holds its upper-left 6x6 quadrant.  Each entry is
 * (float)((double)NF * literal) with NF a float expression (:208-210).
 * ------------------------------------------------------------------------------------------ */
static const double GAUSS_Q[6][6] = {
    {7.76554e-05, 0.000239195, 0.0005738, 0.001072, 0.00155975, 0.00176743},
    {0.000239195, 0.000736774, 0.00176743, 0.00330199, 0.00480437, 0.00544406},
    {0.0005738, 0.00176743, 0.00423984, 0.00792107, 0.0115251, 0.0130596},
    {0.001072, 0.00330199, 0.00792107, 0.0147985, 0.0215317, 0.0243986},
    {0.00155975, 0.00480437, 0.0115251, 0.0215317, 0.0313284, 0.0354998},
    {0.00176743, 0.00544406, 0.0130596, 0.0243986, 0.0354998, 0.0402265},
};

void ora_gaussian_weights(int bits, float w[PATCH][PATCH])
{
    float maxv = bits == 8 ? 255.0f : (bits == 10 ? 1023.0f : 65535.0f);
    volatile float nf = 1.0f / (maxv * maxv * 2.0f * 2.0f);
    for (int i = 0; i < PATCH; i++)
        for (int j = 0; j < PATCH; j++) {
            int qi = i < 6 ? i : 10 - i, qj = j < 6 ? j : 10 - j;
            w[i][j] = (float)((double)nf * GAUSS_Q[qi][qj]);
        }
}

/* ------------------------------------------------------------------------------------------
 * A.3 structure tensor for the pixel at (r, c).  L = LR plane as float, stride W.
 * ------------------------------------------------------------------------------------------ */
static inline float tree11(const float S[PATCH], int odd)
{
    /* sumitup_ps_512 (Raisr_AVX512.cpp:37-44) applied to lanes 1..11 (even pixel of the pair)
     * or lanes 2..12 (odd pixel: weights rotated by shiftR, :111). */
    float u0, u1, u2, u3;
    if (!odd) {
        u0 = S[7] + S[3];
        u1 = (S[0] + S[8]) + S[4];
        u2 = (S[1] + S[9]) + S[5];
        u3 = (S[2] + S[10]) + S[6];
    } else {
        u0 = S[6] + (S[2] + S[10]);
        u1 = S[7] + S[3];
        u2 = (S[0] + S[8]) + S[4];
        u3 = (S[1] + S[9]) + S[5];
    }
    return (u0 + u2) + (u1 + u3);
}

static void gtwg_pixel(const float *L, int W, int r, int c, int odd, const float w[PATCH][PATCH],
                       float *a, float *b, float *d)
{
    float A[PATCH], B[PATCH], D[PATCH];
    for (int k = 0; k < PATCH; k++) A[k] = B[k] = D[k] = 0.0f;
    for (int i = 0; i < PATCH; i++) {
        int y = r - PM + i;
        for (int k = 0; k < PATCH; k++) {
            int x = c - PM + k;
            float gx = L[(size_t)(y + 1) * W + x] - L[(size_t)(y - 1) * W + x];   /* GetGx: row i+2 - row i */
            float gy = L[(size_t)y * W + x + 1] - L[(size_t)y * W + x - 1];       /* GetGy: shiftL - shiftR */
            float p = gx * w[i][k];                                               /* GetGTWG: mul then fma */
            A[k] = fmaf(p, gx, A[k]);
            B[k] = fmaf(p, gy, B[k]);
            float q = gy * w[i][k];
            D[k] = fmaf(q, gy, D[k]);
        }
    }
    *a = tree11(A, odd); *b = tree11(B, odd); *d = tree11(D, odd);
}

/* Same arithmetic as gtwg_pixel() for a run of pixels x in [x0, x1) of row r, with the loops ordered
 * (patch column, patch row, pixel) so that the innermost loop is a unit-stride sweep the compiler
 * vectorises (speed of the CPU baseline only; every lane performs exactly gtwg_pixel's operations in
 * gtwg_pixel's order).  tests/test_oracle_facts.py checks the two forms give identical bits. */
#define GT_BLK 256
static void gtwg_row(const float *L, int W, int r, int x0, int x1, const float w[PATCH][PATCH],
                     float *a, float *b, float *d)
{
    float SA[PATCH][GT_BLK], SB[PATCH][GT_BLK], SD[PATCH][GT_BLK];
    for (int xb = x0; xb < x1; xb += GT_BLK) {
        const int n = (x1 - xb) < GT_BLK ? (x1 - xb) : GT_BLK;
        for (int k = 0; k < PATCH; k++) {
            float *restrict sa = SA[k], *restrict sb = SB[k], *restrict sd = SD[k];
            for (int x = 0; x < n; x++) { sa[x] = 0.0f; sb[x] = 0.0f; sd[x] = 0.0f; }
            for (int i = 0; i < PATCH; i++) {
                const float wv = w[i][k];
                const float *restrict up = L + (size_t)(r - PM + i - 1) * W + (xb - PM + k);
                const float *restrict mid = L + (size_t)(r - PM + i) * W + (xb - PM + k);
                const float *restrict dn = L + (size_t)(r - PM + i + 1) * W + (xb - PM + k);
                for (int x = 0; x < n; x++) {
                    const float gx = dn[x] - up[x];
                    const float gy = mid[x + 1] - mid[x - 1];
                    const float p = gx * wv;
                    sa[x] = fmaf(p, gx, sa[x]);
                    sb[x] = fmaf(p, gy, sb[x]);
                    const float q = gy * wv;
                    sd[x] = fmaf(q, gy, sd[x]);
                }
            }
        }
        for (int x = 0; x < n; x++) {
            float S[PATCH];
            const int odd = (xb + x) & 1;      /* chunk starts are even, so pair parity == column parity */
            for (int k = 0; k < PATCH; k++) S[k] = SA[k][x];
            a[xb + x] = tree11(S, odd);
            for (int k = 0; k < PATCH; k++) S[k] = SB[k][x];
            b[xb + x] = tree11(S, odd);
            for (int k = 0; k < PATCH; k++) S[k] = SD[k][x];
            d[xb + x] = tree11(S, odd);
        }
    }
}

/* exported for the equivalence test */
void ora_gtwg_both(const float *L, int W, int r, int c, int bits, float out6[6])
{
    float w[PATCH][PATCH];
    ora_gaussian_weights(bits, w);
    gtwg_pixel(L, W, r, c, c & 1, w, &out6[0], &out6[1], &out6[2]);
    float a[GT_BLK + 16], b[GT_BLK + 16], d[GT_BLK + 16];
    gtwg_row(L, W, r, c, c + 1, w, a - c + 0, b - c + 0, d - c + 0);
    out6[3] = a[0]; out6[4] = b[0]; out6[5] = d[0];
}

/* ------------------------------------------------------------------------------------------
 * A.4 hash.  x86 cvtps_epi32 semantics: NaN / out of range -> INT_MIN.
 * ------------------------------------------------------------------------------------------ */
static inline int32_t cvt_rne_x86(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int32_t)lrintf(v);      /* default rounding mode = nearest even; v is integral here */
}

static inline float atan2_approx(float y, float x)
{
    /* atan2Approximation_AVX512_32f_16Elements, Raisr_AVX512.cpp:151-173 */
    const float ONEQTR_PI = (float)(M_PI / 4.0);
    const float THRQTR_PI = (float)(3.0 * M_PI / 4.0);
    float abs_y = fabsf(y) + 1e-10f;
    float r1 = (x + abs_y) / (abs_y - x);
    float r2 = (x - abs_y) / (x + abs_y);
    int neg = x < 0.0f;                                     /* _CMP_LT_OQ */
    float rr = neg ? r1 : r2;
    float ang = neg ? THRQTR_PI : ONEQTR_PI;
    ang = fmaf(fmaf(0.1963f * rr, rr, -0.9817f), rr, ang);
    float nang = -1.0f * ang;
    return (y < 0.0f) ? nang : ang;
}

static int hash_pixel(float a, float b, float d, const ora_pass_t *P, int avx2_variant)
{
    const float pi = 3.141592653f;                          /* PI, Raisr_globals.h:29 */
    float T = a + d;
    float Dt = (a * d) - (b * b);
    float rad = ((T * T) / 4.0f) - Dt;
    float s = avx2_variant ? x86_rcp(x86_rsqrt(rad)) : x86_rcp14(x86_rsqrt14(rad));
    float hT = T / 2.0f;
    float L1 = hT + s;
    float L2 = hT - s;
    float xx = (b < 0.0f || b > 0.0f) ? (L1 - d) : 1.0f;    /* _CMP_NEQ_OQ: ordered, NaN -> 1 */
    float ang = atan2_approx(b, xx);
    ang = ang + ((ang < 0.0f) ? pi : 0.0f);
    float sL1 = avx2_variant ? x86_rcp(x86_rsqrt(L1)) : x86_rcp14(x86_rsqrt14(L1));
    float sL2 = avx2_variant ? x86_rcp(x86_rsqrt(L2)) : x86_rcp14(x86_rsqrt14(L2));
    float coh = (sL1 - sL2) / ((sL1 + sL2) + 1e-17f);
    float str = L1;
    int32_t ai = cvt_rne_x86(floorf(ang * P->qangle));
    if (ai < 0) ai = 0;
    if (ai > 23) ai = 23;
    int si, ci;
    if (!avx2_variant) {
        /* Raisr_AVX512.cpp:242-249: [Q <= v], ordered => NaN counts 0 */
        si = (P->qstr[0] <= str) + (P->qstr[1] <= str);
        ci = (P->qcoh[0] <= coh) + (P->qcoh[1] <= coh);
    } else {
        /* Raisr_AVX256.cpp:457-464: 2 - [v <= Q0] - [v <= Q1], NaN => 2 */
        si = 2 - ((str <= P->qstr[0]) + (str <= P->qstr[1]));
        ci = 2 - ((coh <= P->qcoh[0]) + (coh <= P->qcoh[1]));
    }
    return ai * 9 + si * 3 + ci;
}

/* ------------------------------------------------------------------------------------------
 * A.5 filter application.
 * ------------------------------------------------------------------------------------------ */
static float dot_patch(const float *L, int W, int r, int c, const float *f)
{
    /* pixbuf[128] / filter row padded to 128 with +0 (Raisr.cpp:1056, :329-331).  The patch is gathered row by row (11
     * contiguous floats each); the filter row is used in place for the seven full chunks and copied only for the last,
     * partly padded one.  Same 16 accumulator chains, same order: acc[l] = p[l]*f[l]; acc[l] = fma(p[16c+l], f[16c+l], acc[l]). */
    float pb[128] __attribute__((aligned(64))), ft[16] __attribute__((aligned(64))), acc[16] __attribute__((aligned(64)));
    for (int i = 0; i < PATCH; i++) memcpy(pb + i * PATCH, L + (size_t)(r - PM + i) * W + (c - PM), PATCH * sizeof(float));
    for (int k = TAPS; k < 128; k++) pb[k] = 0.0f;
    for (int l = 0; l < 16; l++) ft[l] = (112 + l < TAPS) ? f[112 + l] : 0.0f;
    for (int l = 0; l < 16; l++) acc[l] = pb[l] * f[l];
    for (int ch = 1; ch < 7; ch++)
        for (int l = 0; l < 16; l++) acc[l] = fmaf(pb[16 * ch + l], f[16 * ch + l], acc[l]);
    for (int l = 0; l < 16; l++) acc[l] = fmaf(pb[112 + l], ft[l], acc[l]);
    float t[8], u[4];
    for (int i = 0; i < 8; i++) t[i] = acc[i] + acc[i + 8];
    for (int i = 0; i < 4; i++) u[i] = t[i] + t[i + 4];
    return (u[0] + u[2]) + (u[1] + u[3]);
}

/* ------------------------------------------------------------------------------------------
 * One RAISR pass over an integer LR plane (already at this pass's resolution).
 * out must be preset by the caller where the reference leaves caller memory untouched
 * (Randomness blending only).  Optional dumps: hash1 (first hash per pixel, -1 where
 * not filtered), hr (float HR plane).
 * ------------------------------------------------------------------------------------------ */
void ora_pass(const uint16_t *lr, int W, int H, const ora_pass_t *P, uint16_t *out,
              int32_t *hash_dump, float *hr_dump)
{
    float wg[PATCH][PATCH];
    ora_gaussian_weights(P->bits, wg);
    size_t n = (size_t)W * H;
    float *L = (float *)malloc(n * sizeof(float));
    float *HR = (float *)malloc(n * sizeof(float));
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) { L[i] = (float)lr[i]; HR[i] = L[i]; }   /* ippiConvert_*32f (exact); HR := LR, Raisr.cpp:1035 */
    if (hash_dump) for (size_t i = 0; i < n; i++) hash_dump[i] = -1;
    const float lo = (float)P->lo, hi = (float)P->hi;
    const int randomness = P->blending == ORA_BLEND_RANDOMNESS;

    if (randomness) {
        /* every pixel the loop below does not write keeps the unclamped LR copy, except the
         * never-written pixels [c_final, W-6) of row H-7 (left as the caller preset them). */
        int c_end = LM;
        {   /* replay the column driver once to learn the final c */
            int loopItr = P->asm_type == ORA_ASM_AVX512 ? 16 : 8, unroll = loopItr, c = LM;
            while (c + loopItr <= W - LM) {
                if (loopItr > 8 && c + 2 * unroll > W - LM) loopItr = 8;
                c += loopItr;
            }
            c_end = c;
        }
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int untouched = (H >= 2 * LM + 1) && r == H - LM - 1 && c >= c_end && c < W - LM;
                if (!untouched) out[(size_t)r * W + c] = lr[(size_t)r * W + c];
            }
    }

    #pragma omp parallel for schedule(dynamic, 2)
    for (int r = LM; r < H - LM; r++) {                            /* Raisr.cpp:1036-1058 */
        float *ga = (float *)malloc(sizeof(float) * 3 * (size_t)W), *gb = ga + W, *gd = gb + W;
        if (W > 2 * LM) gtwg_row(L, W, r, LM, W - LM, wg, ga, gb, gd);
        int unroll = P->asm_type == ORA_ASM_AVX512 ? 16 : 8;       /* unrollSizePatchBased, :1481-1528 */
        int loopItr = unroll;
        int c = LM;
        while (c + loopItr <= W - LM) {                            /* :1066 */
            int avx2_hash = (loopItr == 8);                        /* :1133-1141 */
            for (int pix = 0; pix < loopItr; pix++) {
                int cc = c + pix;
                int odd = pix & 1;                                 /* pair position inside computeGTWG call */
                (void)odd;
                const float a = ga[cc], b = gb[cc], d = gd[cc];   /* == gtwg_pixel(L, W, r, cc, odd, ...) */
                int h = hash_pixel(a, b, d, P, avx2_hash);
                int t = 0;
                if (P->pixel_types == 4) t = ((r - PM) % 2) * 2 + ((cc - PM) % 2);   /* :1068-1096 */
                const float *f = P->bank + ((size_t)h * P->pixel_types + t) * TAPS;
                float v = dot_patch(L, W, r, cc, f);
                size_t idx = (size_t)r * W + cc;
                if (hash_dump && hash_dump[idx] < 0) hash_dump[idx] = h;
                float cur;
                if (v > lo && v < hi) { HR[idx] = v; cur = v; }    /* :1196-1200 */
                else cur = L[idx];
                if (randomness) {                                  /* :1203-1242 */
                    int census = 0;
                    for (int i = -1; i <= 1; i++)
                        for (int j = -1; j <= 1; j++)
                            if (i || j) census += L[(size_t)(r + i) * W + cc + j] < L[idx];
                    float weight = (float)census / 8.0f;
                    float val = weight * cur + (1.0f - weight) * L[idx];
                    val = (float)((double)val + 0.5);
                    float cl = val < lo ? lo : (val > hi ? hi : val);
                    out[idx] = (uint16_t)cl;                       /* C truncation */
                }
            }
            if (loopItr > 8 && c + 2 * unroll > W - LM) loopItr = 8;   /* :1246-1249 */
            c += loopItr;
        }
        free(ga);
    }

    if (!randomness) {
        /* borders = unclamped integer LR (Raisr.cpp:999-1028,1252-1265); interior = blend stage */
        for (int c = 0; c < W; c++) { out[c] = lr[c]; out[(size_t)(H - 1) * W + c] = lr[(size_t)(H - 1) * W + c]; }
        for (int r = 0; r < H; r++) { out[(size_t)r * W] = lr[(size_t)r * W]; out[(size_t)r * W + W - 1] = lr[(size_t)r * W + W - 1]; }
        #pragma omp parallel for schedule(static)
        for (int r = 1; r < H - 1; r++) {                          /* Raisr_AVX256.cpp:78-165 */
            /* the eight neighbours written out, so that the column loop is a flat loop the compiler vectorises */
            const float *l0 = L + (size_t)(r - 1) * W, *l1 = L + (size_t)r * W, *l2 = L + (size_t)(r + 1) * W;
            const float *h0 = HR + (size_t)(r - 1) * W, *h1 = HR + (size_t)r * W, *h2 = HR + (size_t)(r + 1) * W;
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
                float weight = (float)hd / 8.0f;
                float w2 = 1.0f - weight;
                float val = (weight * lc) + (w2 * hc);
                val = val + 0.5f;
                int32_t iv = cvt_rne_x86(floorf(val));
                if (iv > ihi) iv = ihi;
                if (iv < ilo) iv = ilo;
                o[c] = (uint16_t)iv;
            }
        }
    }
    if (hr_dump) memcpy(hr_dump, HR, n * sizeof(float));
    free(L); free(HR);
}

/* ------------------------------------------------------------------------------------------
 * A.7 whole Y-plane job.  passes in {1,2}; mode in {1,2} (RNLInit twoPassMode).
 * in/out are value planes (uint16 containers, tightly packed).
 * ------------------------------------------------------------------------------------------ */
void ora_process_y(const uint16_t *in, int inW, int inH, uint16_t *out, int outW, int outH,
                   int passes, int mode, const ora_pass_t *P1, const ora_pass_t *P2, int tie)
{
    if (passes == 1) {
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        ora_resize_bilinear(in, inW, inH, inW, lr, outW, outH, outW, tie);
        ora_pass(lr, outW, outH, P1, out, NULL, NULL);
        free(lr);
        return;
    }
    if (mode == 2) {
        /* pass 1 at input size without upscale; pass 2 upscales the intermediate (Raisr.cpp:945-975) */
        uint16_t *mid = (uint16_t *)calloc((size_t)inW * inH, sizeof(uint16_t));
        ora_pass(in, inW, inH, P1, mid, NULL, NULL);
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        ora_resize_bilinear(mid, inW, inH, inW, lr, outW, outH, outW, tie);
        ora_pass(lr, outW, outH, P2, out, NULL, NULL);
        free(mid); free(lr);
    } else {
        uint16_t *lr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)outW * outH);
        uint16_t *mid = (uint16_t *)calloc((size_t)outW * outH, sizeof(uint16_t));
        ora_resize_bilinear(in, inW, inH, inW, lr, outW, outH, outW, tie);
        ora_pass(lr, outW, outH, P1, mid, NULL, NULL);
        ora_pass(mid, outW, outH, P2, out, NULL, NULL);
        free(lr); free(mid);
    }
}

/* exported scalar helpers so tests can probe single stages */
float ora_x86_rcp14(float x) { return x86_rcp14(x); }
float ora_x86_rsqrt14(float x) { return x86_rsqrt14(x); }
float ora_x86_rcp(float x) { return x86_rcp(x); }
float ora_x86_rsqrt(float x) { return x86_rsqrt(x); }
int ora_hash(float a, float b, float d, const ora_pass_t *P, int avx2_variant) { return hash_pixel(a, b, d, P, avx2_variant); }
/* n (a, b, d) triples -> n hash buckets */
void ora_hash_array(const float *abd, size_t n, const ora_pass_t *P, int avx2_variant, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)hash_pixel(abd[3 * i], abd[3 * i + 1], abd[3 * i + 2], P, avx2_variant);
}

/* Introspection for tests: the exact structure tensor (a, b, d) of every pixel of the filtered rows/columns
 * [LM, H-LM) x [LM, W-LM) of an integer LR plane, exactly as ora_pass computes it (gtwg_row).  Planes a, b, d are
 * W x H floats; pixels outside the zone are left untouched. */
void ora_tensor_plane(const uint16_t *lr, int W, int H, int bits, float *a, float *b, float *d)
{
    float wg[PATCH][PATCH];
    ora_gaussian_weights(bits, wg);
    size_t n = (size_t)W * H;
    float *L = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) L[i] = (float)lr[i];
    #pragma omp parallel for schedule(dynamic, 2)
    for (int r = LM; r < H - LM; r++)
        if (W > 2 * LM) gtwg_row(L, W, r, LM, W - LM, wg, a + (size_t)r * W, b + (size_t)r * W, d + (size_t)r * W);
    free(L);
}
