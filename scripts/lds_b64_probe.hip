// lds_b64_probe.hip -- does gfx950 serve a ds_read_b64 whose address is only 4-byte aligned, and at what rate?
// The filter stage of k_hashfilter_ac fetches one tap of TWO pixel steps with one ds_read2_b32 (two 4-byte accesses 16 B apart: 4 LDS
// cycles per wave-instruction, MI355X_MICROARCH.md).  If the two pixels of a lane are ADJACENT columns instead, the same two window
// values are one ds_read_b64 (2 LDS cycles when conflict-free) -- but the window's row stride is odd (77 dwords), so half of those
// addresses are 4 mod 8.  The compiler refuses to emit ds_read_b64 for such an address (it emits ds_read2_b32 offset1:1); the kernel
// driver runs gfx9 queues with SH_MEM_CONFIG.alignment_mode = unaligned, so the hardware may well accept it.  This probe answers:
//   (1) correctness: ds_read_b64 at byte address 4 (3 lane + o), o = 0 / 1, against the known LDS contents;
//   (2) rate: cycles per wave-instruction and CU for  b32 | read2_b32 (the production pattern) | b64 aligned | b64 at 4 mod 8 |
//       b64 in the filter stage's own address pattern (tap k = 16 ch + l of the 11 x 11 window, row stride 77, lane group g at columns
//       2 g, 2 g + 1) | read2_b32 in the production pattern (group g at columns g, g + 4).
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_b64_probe.hip -o /tmp/lds_b64_probe && /tmp/lds_b64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(256) void k_correct(uint32_t* out, int o)
{
    __shared__ uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0xA0000000u + i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(s) + 4u * (threadIdx.x * 3u + (uint32_t)o);
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[2 * threadIdx.x] = (uint32_t)v;
    out[2 * threadIdx.x + 1] = (uint32_t)(v >> 32);
}

// MODE 0: ds_read_b32 linear   1: ds_read2_b32 linear (offset1:4)   2: ds_read_b64 aligned linear   3: ds_read_b64 at 4 mod 8 linear
//      4: ds_read_b64, filter pattern (new pixel order)   5: ds_read2_b32 offset1:4, filter pattern (production pixel order)
//      6: ds_read_b64, filter pattern on an EVEN row stride (78) -- alignment then depends on the tap column only
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_rate(uint32_t* out, int iters)
{
    __shared__ uint32_t s[28 * 78 + 64];
    for (int i = threadIdx.x; i < 28 * 78 + 64; i += 256) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15, w = threadIdx.x >> 6;
    uint32_t a[8];
    const uint32_t base = (uint32_t)(uintptr_t)(s);
    for (int ch = 0; ch < 8; ch++) {
        const int k = 16 * ch + l;
        const int LW = MODE == 6 ? 78 : 77;
        const int pos = (k < 121) ? (k / 11) * LW + (k % 11) : 0;
        if (MODE == 0 || MODE == 1) a[ch] = base + 4u * (uint32_t)(lane + 65 * ch);
        if (MODE == 2) a[ch] = base + 8u * (uint32_t)(lane + 33 * ch);
        if (MODE == 3) a[ch] = base + 8u * (uint32_t)(lane + 33 * ch) + 4u;
        if (MODE == 4 || MODE == 6) a[ch] = base + 4u * (uint32_t)(4 * w * LW + pos + 2 * g);
        if (MODE == 5) a[ch] = base + 4u * (uint32_t)(4 * w * LW + pos + g);
    }
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t v[8];
#pragma unroll
        for (int ch = 0; ch < 8; ch++) {
            if (MODE == 0) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(a[ch])); v[ch] = t; }
            else if (MODE == 1 || MODE == 5) asm volatile("ds_read2_b32 %0, %1 offset1:4" : "=v"(v[ch]) : "v"(a[ch]));
            else asm volatile("ds_read_b64 %0, %1" : "=v"(v[ch]) : "v"(a[ch]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ch = 0; ch < 8; ch++) acc ^= v[ch];
#pragma unroll
        for (int ch = 0; ch < 8; ch++) asm volatile("" : "+v"(a[ch]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <int MODE>
void run(const char* name)
{
    const int blocks = 256 * 4, iters = 20000;
    uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_rate<MODE><<<blocks, 256>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_rate<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst_per_cu = 16.0 /*waves*/ * iters * 8;
    printf("%-58s %8.3f ms   %.2f cycles per wave-instruction and CU @2.4 GHz\n", name, ms, 2.4e9 * ms * 1e-3 / inst_per_cu);
    hipFree(d);
}

int main()
{
    uint32_t* d; hipMalloc(&d, 512 * 4);
    std::vector<uint32_t> h(512);
    for (int o = 0; o < 2; o++) {
        k_correct<<<1, 256>>>(d, o);
        hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; t++) {
            const uint32_t e0 = 0xA0000000u + 3 * t + o;
            if (h[2 * t] != e0 || h[2 * t + 1] != e0 + 1) { if (bad < 4) printf("  lane %d: got %08x %08x want %08x %08x\n", t, h[2 * t], h[2 * t + 1], e0, e0 + 1); bad++; }
        }
        printf("correctness, dword offset 3 lane + %d (lanes alternate 0 / 4 mod 8): %d of 256 lanes wrong\n", o, bad);
    }
    run<0>("ds_read_b32 linear");
    run<1>("ds_read2_b32 offset1:4 linear");
    run<2>("ds_read_b64 8-byte aligned linear");
    run<3>("ds_read_b64 at 4 mod 8 linear");
    run<4>("ds_read_b64 filter pattern, stride 77 (columns 2g, 2g+1)");
    run<5>("ds_read2_b32 filter pattern, stride 77 (production: g, g+4)");
    run<6>("ds_read_b64 filter pattern, stride 78");
    return 0;
}
