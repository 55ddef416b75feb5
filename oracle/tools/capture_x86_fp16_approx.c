/*
 * capture_x86_fp16_approx.c -- TEST INFRASTRUCTURE (oracle tooling), not product code.
 *
 * Dumps VRCPPH and VRSQRTPH (the fp16 approximation instructions of the reference's AVX512-FP16
 * hash, Library/Raisr_AVX512FP16.cpp:412,436-437,522,551-552) for ALL 65536 binary16 inputs on the
 * CPU it runs on (needs AVX512-FP16: Sapphire Rapids or later), plus a self-check of the basic
 * binary16 arithmetic this CPU performs (add/mul/div/fma on random operands) that the oracle's
 * software fp16 model is validated against.
 *
 * build: /opt/rocm/lib/llvm/bin/clang -O2 -mavx512fp16 -mavx512vl capture_x86_fp16_approx.c -o cap16
 * usage: ./cap16 <outdir>     -> rcpph.bin, rsqrtph.bin (65536 x u16 each), fp16_arith.bin
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline _Float16 u2h(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return h; }
static inline uint16_t h2u(_Float16 h) { uint16_t u; memcpy(&u, &h, 2); return u; }

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    char path[1024];
    uint16_t *o = (uint16_t *)malloc(65536 * 2);
    for (uint32_t i = 0; i < 65536; i++) { __m128h x = _mm_set_sh(u2h((uint16_t)i)); o[i] = h2u(_mm_cvtsh_h(_mm_rcp_sh(x, x))); }
    snprintf(path, sizeof path, "%s/rcpph.bin", dir);
    FILE *f = fopen(path, "wb"); fwrite(o, 2, 65536, f); fclose(f);
    for (uint32_t i = 0; i < 65536; i++) { __m128h x = _mm_set_sh(u2h((uint16_t)i)); o[i] = h2u(_mm_cvtsh_h(_mm_rsqrt_sh(x, x))); }
    snprintf(path, sizeof path, "%s/rsqrtph.bin", dir);
    f = fopen(path, "wb"); fwrite(o, 2, 65536, f); fclose(f);

    /* arithmetic vectors: records of 7 u16: a b c  a+b  a*b  a/b  fma(a,b,c) */
    snprintf(path, sizeof path, "%s/fp16_arith.bin", dir);
    f = fopen(path, "wb");
    uint32_t lcg = 2024u;
    for (int i = 0; i < 200000; i++) {
        uint16_t v[3];
        for (int k = 0; k < 3; k++) {
            lcg = lcg * 1664525u + 1013904223u;
            v[k] = (uint16_t)(lcg >> 16);
            if ((i & 3) == 1) v[k] &= 0x83ffu | (uint16_t)((lcg >> 3) & 0x3c00u);   /* favour small exponents / subnormals */
        }
        volatile _Float16 a = u2h(v[0]), b = u2h(v[1]), c = u2h(v[2]);
        __m128h A = _mm_set_sh(a), B = _mm_set_sh(b), C = _mm_set_sh(c);
        uint16_t r[7] = { v[0], v[1], v[2], h2u(_mm_cvtsh_h(_mm_add_sh(A, B))), h2u(_mm_cvtsh_h(_mm_mul_sh(A, B))),
                          h2u(_mm_cvtsh_h(_mm_div_sh(A, B))), h2u(_mm_cvtsh_h(_mm_fmadd_sh(A, B, C))) };
        fwrite(r, 2, 7, f);
    }
    fclose(f);
    return 0;
}
