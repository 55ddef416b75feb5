/*
 * capture_x86_approx.c -- TEST INFRASTRUCTURE (oracle tooling), not product code.
 *
 * Dumps the exhaustive behaviour of the four x86 approximation instructions the
 * reference's hash stage executes, on the CPU this program runs on:
 *   VRCP14SS / VRSQRT14SS   (reference: Library/Raisr_AVX512.cpp:200,221-222)
 *   RCPSS    / RSQRTSS      (reference: Library/Raisr_AVX256.cpp:412,436-437)
 * One output file per instruction and exponent parity, 2^23 little-endian u32
 * result bit patterns each (input mantissa = index, exponent 127 or 128).
 * A second section prints special-value results (zero, inf, NaN, negative,
 * denormal, extreme exponents) and an exponent-independence sweep.
 *
 * Must run on a GenuineIntel CPU with AVX-512F: RCPSS/RSQRTSS are not
 * architecturally specified and differ between vendors; the reference's published
 * numbers and its "AVX-512 path" are Intel's.  fit_x86_approx.py turns the dumps
 * into the compact exact models that oracle/ and the HIP kernels embed.
 *
 * build: gcc -O2 -mavx512f -mavx512vl capture_x86_approx.c -o capture_x86_approx
 * usage: ./capture_x86_approx <outdir>
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static uint32_t do_rcp14(uint32_t b)   { __m128 x = _mm_set_ss(u2f(b)); return f2u(_mm_cvtss_f32(_mm_rcp14_ss(x, x))); }
static uint32_t do_rsqrt14(uint32_t b) { __m128 x = _mm_set_ss(u2f(b)); return f2u(_mm_cvtss_f32(_mm_rsqrt14_ss(x, x))); }
static uint32_t do_rcp(uint32_t b)     { __m128 x = _mm_set_ss(u2f(b)); return f2u(_mm_cvtss_f32(_mm_rcp_ss(x))); }
static uint32_t do_rsqrt(uint32_t b)   { __m128 x = _mm_set_ss(u2f(b)); return f2u(_mm_cvtss_f32(_mm_rsqrt_ss(x))); }

typedef uint32_t (*fn_t)(uint32_t);

static void dump(const char *dir, const char *name, fn_t fn, uint32_t expbits)
{
    const uint32_t n = 1u << 23;
    uint32_t *o = (uint32_t *)malloc((size_t)n * 4);
    char path[1024];
    for (uint32_t m = 0; m < n; m++) o[m] = fn(expbits | m);
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); exit(1); }
    fwrite(o, 4, n, f);
    fclose(f);
    free(o);
}

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    dump(dir, "rcp14", do_rcp14, 0x3f800000u);
    dump(dir, "rsqrt14_e0", do_rsqrt14, 0x3f800000u);
    dump(dir, "rsqrt14_e1", do_rsqrt14, 0x40000000u);
    dump(dir, "rcp", do_rcp, 0x3f800000u);
    dump(dir, "rsqrt_e0", do_rsqrt, 0x3f800000u);
    dump(dir, "rsqrt_e1", do_rsqrt, 0x40000000u);

    /* special values + exponent sweep: "in rcp14 rsqrt14 rcp rsqrt" hex words */
    char path[1024];
    snprintf(path, sizeof path, "%s/special.txt", dir);
    FILE *f = fopen(path, "w");
    static const uint32_t sp[] = {
        0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x7f800001u,
        0x00000001u, 0x007fffffu, 0x00400000u, 0x00200000u, 0x00000100u, 0x00800000u,
        0x00800001u, 0x7f7fffffu, 0x7f000000u, 0x7e800000u, 0x7e800001u, 0x7f000001u,
        0xbf800000u, 0x80000001u, 0x3f800000u, 0x3f800001u, 0x40000000u, 0x3f000000u,
        0x7e7fffffu, 0x7effffffu, 0xc0490fdbu, 0x80800000u };
    for (unsigned i = 0; i < sizeof sp / sizeof sp[0]; i++)
        fprintf(f, "%08x %08x %08x %08x %08x\n", sp[i], do_rcp14(sp[i]), do_rsqrt14(sp[i]), do_rcp(sp[i]), do_rsqrt(sp[i]));
    /* every normal exponent x a spread of mantissas */
    uint32_t lcg = 12345u;
    for (uint32_t e = 1; e < 255; e++)
        for (int k = 0; k < 12; k++) {
            lcg = lcg * 1664525u + 1013904223u;
            uint32_t b = (e << 23) | (lcg >> 9);
            fprintf(f, "%08x %08x %08x %08x %08x\n", b, do_rcp14(b), do_rsqrt14(b), do_rcp(b), do_rsqrt(b));
        }
    fclose(f);
    return 0;
}
