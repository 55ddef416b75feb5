// fma_mix_probe.hip -- issue rate of v_fma_mix_f32 (binary16 operand widened inside the FMA: exact) against v_fma_f32 on gfx950, and
// whether the compiler selects it for fmaf((float)h, q, acc).  Decides whether the filter stage could keep its window samples as packed
// binary16 pairs (8- and 10-bit samples are exact there) and feed them to the fp32 chains without conversion instructions.
//   hipcc --offload-arch=gfx950 -O3 scripts/fma_mix_probe.hip -o /tmp/fma_mix_probe && /tmp/fma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float s, unsigned hx)
{
    float a[8];
    h2 x[4];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    for (int i = 0; i < 4; i++) x[i] = __builtin_bit_cast(h2, hx + (unsigned)i * 0x00010001u + threadIdx.x);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x[i >> 1]), "v"(s));
                if (MODE == 1) {                                                                                    // lo / hi half of the packed pair
                    if (i & 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(x[i >> 1]), "v"(s));
                    else       asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(x[i >> 1]), "v"(s));
                }
                if (MODE == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x[i >> 1]), "v"(s));
            }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 4096;
    float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(d, 16, 0.999f, 0x3c003c00u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0x3c003c00u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double inst = (double)blocks * 4 * iters * 32;
    printf("%-34s %d WG/CU %8.3f ms  cycles per wave-inst and SIMD @2.4 GHz: %.2f\n", name, blocks_per_cu, ms, 2.4e9 * ms * 1e-3 * 1024 / inst);
    (void)hipFree(d);
}

int main()
{
    for (int wg = 4; wg <= 8; wg += 4) {
        run<0>("v_fma_f32", wg);
        run<1>("v_fma_mix_f32 (f16 lo/hi, f32, f32)", wg);
        run<2>("v_fmac_f32", wg);
    }
    return 0;
}
