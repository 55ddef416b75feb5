// x86_approx_dev.h -- device-side bit-exact models of the x86 approximation instructions the
// reference's hash stage executes (VRCP14PS/VRSQRT14PS: reference Library/Raisr_AVX512.cpp:200,
// 221-222; RCPPS/RSQRTPS: Library/Raisr_AVX256.cpp:412,436-437).  Coefficient tables come from
// x86_approx_tables.h (generated from an exhaustive sweep of a GenuineIntel AVX-512 core).
//
// Layout on the device: one 64-entry uint2 {C0, C1} table per instruction (staged into LDS by
// the hash kernel: the index is data dependent per lane) and the two legacy LUTs in global memory
// (touched only by the few tail columns that replay the AVX2 hash).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace x86dev {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// significand/exponent packing with truncating denormalisation (observed VRCP14 behaviour)
__device__ __forceinline__ uint32_t pack_trunc(uint32_t sign, int e, uint32_t sig)
{
    if (e >= 255) return sign | 0x7f800000u;
    if (e >= 1) return sign | ((uint32_t)e << 23) | (sig & 0x7fffffu);
    int sh = 1 - e;
    if (sh > 24) return sign;
    return sign | (sig >> sh);
}

__device__ __forceinline__ int norm_denormal(uint32_t frac, uint32_t& m)
{
    int lz = __clz((int)frac) - 8;          // shifts needed to bring the leading one to bit 23
    m = (frac << lz) & 0x7fffffu;
    return 1 - lz;
}

// VRCP14: tab[i] = {C0, C1}
__device__ __forceinline__ float rcp14(float xf, const uint2* __restrict__ tab)
{
    uint32_t x = f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255) return m ? u2f(x | 0x00400000u) : u2f(sign);
    if (E == 0) {
        if (m == 0) return u2f(sign | 0x7f800000u);
        E = norm_denormal(m, m);
    }
    if (m == 0) return u2f(pack_trunc(sign, 254 - E, 0x800000u));
    const uint2 c = tab[m >> 17];
    const uint32_t t = (m >> 7) & 1023u;
    const uint32_t code = (c.x - c.y * t) >> 9;
    return u2f(pack_trunc(sign, 253 - E, 0x800000u | (code << 7)));
}

// VRSQRT14: tab[32*parity + i] = {C0, C1}
__device__ __forceinline__ float rsqrt14(float xf, const uint2* __restrict__ tab)
{
    uint32_t x = f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255 && m) return u2f(x | 0x00400000u);
    if (E == 0 && m == 0) return u2f(sign | 0x7f800000u);
    if (sign) return u2f(0xffc00000u);
    if (E == 255) return 0.0f;
    if (E == 0) E = norm_denormal(m, m);
    const int ue = E - 127;
    const int p = ue & 1;
    const int half = (ue - p) >> 1;            // exact: ue - p is even
    if (p == 0 && m == 0) return u2f((uint32_t)(127 - half) << 23);
    const uint2 c = tab[32 * p + (m >> 18)];
    const uint32_t t = (m >> 8) & 1023u;
    const uint32_t code = (c.x - c.y * t) >> 9;
    return u2f(((uint32_t)(126 - half) << 23) | (code << 7));
}

// RCPPS (legacy): lut[2048]
__device__ __forceinline__ float rcp_legacy(float xf, const uint16_t* __restrict__ lut)
{
    uint32_t x = f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255) return m ? u2f(x | 0x00400000u) : u2f(sign);
    if (E == 0) return u2f(sign | 0x7f800000u);
    int re = 253 - E;
    if (re <= 0) return u2f(sign);
    return u2f(sign | ((uint32_t)re << 23) | ((uint32_t)lut[m >> 12] << 11));
}

// RSQRTPS (legacy): lut[2][1024]
__device__ __forceinline__ float rsqrt_legacy(float xf, const uint16_t* __restrict__ lut)
{
    uint32_t x = f2u(xf), sign = x & 0x80000000u;
    int E = (int)((x >> 23) & 0xff);
    uint32_t m = x & 0x7fffffu;
    if (E == 255 && m) return u2f(x | 0x00400000u);
    if (E == 0) return u2f(sign | 0x7f800000u);
    if (sign) return u2f(0xffc00000u);
    if (E == 255) return 0.0f;
    const int ue = E - 127;
    const int p = ue & 1;
    const int half = (ue - p) >> 1;
    return u2f(((uint32_t)(126 - half) << 23) | ((uint32_t)lut[1024 * p + (m >> 13)] << 11));
}

}  // namespace x86dev
