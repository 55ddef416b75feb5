/*
 * x86_approx.h -- TEST INFRASTRUCTURE (part of oracle/; never linked into the product).
 *
 * Portable bit-exact software models of the x86 approximation instructions that decide
 * bucket boundaries in the reference's hash stage:
 *   VRCP14PS / VRSQRT14PS  -- reference Library/Raisr_AVX512.cpp:200,221-222
 *   RCPPS    / RSQRTPS     -- reference Library/Raisr_AVX256.cpp:412,436-437 (also executed on the
 *                             AVX-512 path for the tail columns, Library/Raisr.cpp:1133-1135)
 * Coefficients/LUTs live in x86_approx_tables.h (generated; exhaustively verified against a
 * GenuineIntel AVX-512 core by oracle/tools/fit_x86_approx.py).  Special-value behaviour follows
 * the captured results in tests/golden/x86_approx_special.txt.
 */
#pragma once
#include <stdint.h>
#include <string.h>
#include "x86_approx_tables.h"

static inline uint32_t x86a_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float x86a_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Build a float from sign, unbiased-by-nothing "biased exponent" e (may be <= 0 or >= 255) and a
 * 24-bit significand sig (bit 23 = implicit one).  Results below the normal range are produced by
 * truncating right shifts (observed VRCP14 behaviour: 0x7e800001 -> 0x007fff00). */
static inline uint32_t x86a_pack_trunc(uint32_t sign, int e, uint32_t sig)
{
    if (e >= 255) return sign | 0x7f800000u;
    if (e >= 1) return sign | ((uint32_t)e << 23) | (sig & 0x7fffffu);
    int sh = 1 - e;
    if (sh > 24) return sign;
    return sign | (sig >> sh);
}

/* normalise a denormal: returns biased exponent (<= 0) and sets *m to the 23-bit mantissa */
static inline int x86a_norm_denormal(uint32_t frac, uint32_t *m)
{
    int e = 1;
    while (!(frac & 0x800000u)) { frac <<= 1; e--; }
    *m = frac & 0x7fffffu;
    return e;
}

/* VRCP14SS */
static inline float x86_rcp14(float xf)
{
    uint32_t x = x86a_f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255) return m ? x86a_u2f(x | 0x00400000u) : x86a_u2f(sign);       /* NaN -> QNaN, inf -> 0 */
    if (E == 0) {
        if (m == 0) return x86a_u2f(sign | 0x7f800000u);                          /* 1/0 = inf */
        E = x86a_norm_denormal(m, &m);                                            /* DAZ off: normalise */
    }
    if (m == 0) return x86a_u2f(x86a_pack_trunc(sign, 254 - E, 0x800000u));
    uint32_t i = m >> 17, t = (m >> 7) & 1023u;
    uint32_t code = (X86_RCP14_C0[i] - (uint32_t)X86_RCP14_C1[i] * t) >> 9;
    return x86a_u2f(x86a_pack_trunc(sign, 253 - E, 0x800000u | (code << 7)));
}

/* VRSQRT14SS */
static inline float x86_rsqrt14(float xf)
{
    uint32_t x = x86a_f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255 && m) return x86a_u2f(x | 0x00400000u);                         /* NaN */
    if (E == 0 && m == 0) return x86a_u2f(sign | 0x7f800000u);                   /* +-0 -> +-inf */
    if (sign) return x86a_u2f(0xffc00000u);                                       /* negative -> QNaN indefinite */
    if (E == 255) return 0.0f;                                                    /* +inf -> +0 */
    if (E == 0) E = x86a_norm_denormal(m, &m);
    int ue = E - 127;
    int p = ue & 1;                         /* parity (works for negative ue in two's complement) */
    int half = (ue - p) / 2;                /* exact */
    if (p == 0 && m == 0) return x86a_u2f((uint32_t)(127 - half) << 23);
    uint32_t i = m >> 18, t = (m >> 8) & 1023u;
    uint32_t code = (X86_RSQRT14_C0[32 * p + i] - (uint32_t)X86_RSQRT14_C1[32 * p + i] * t) >> 9;
    return x86a_u2f(((uint32_t)(126 - half) << 23) | (code << 7));
}

/* RCPSS (legacy 12-bit): denormal inputs behave as zero, results below the normal range flush to 0 */
static inline float x86_rcp(float xf)
{
    uint32_t x = x86a_f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255) return m ? x86a_u2f(x | 0x00400000u) : x86a_u2f(sign);
    if (E == 0) return x86a_u2f(sign | 0x7f800000u);
    int re = 253 - E;
    if (re <= 0) return x86a_u2f(sign);
    return x86a_u2f(sign | ((uint32_t)re << 23) | ((uint32_t)X86_RCP_LUT[m >> 12] << 11));
}

/* RSQRTSS (legacy 12-bit) */
static inline float x86_rsqrt(float xf)
{
    uint32_t x = x86a_f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255 && m) return x86a_u2f(x | 0x00400000u);
    if (E == 0) return x86a_u2f(sign | 0x7f800000u);                              /* zero and denormals */
    if (sign) return x86a_u2f(0xffc00000u);
    if (E == 255) return 0.0f;
    int ue = E - 127;
    int p = ue & 1;
    int half = (ue - p) / 2;
    return x86a_u2f(((uint32_t)(126 - half) << 23) | ((uint32_t)X86_RSQRT_LUT[1024 * p + (m >> 13)] << 11));
}
