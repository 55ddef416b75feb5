// valu_rate_probe.hip -- measures issue rates of v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 on gfx950
// (decides whether the k_hash inner loop should use packed fp32 math).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float s)
{
    float a[8]; f2 v[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; v[i] = (f2){a[i], a[i] + 0.5f}; }
    const f2 s2 = {s, s * 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.25f);
                if (MODE == 1) v[i] = __builtin_elementwise_fma(v[i], s2, s2);
                if (MODE == 2) v[i] = v[i] * s2;
            }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
double run(const char* name, int flop_per_inst)
{
    const int blocks = 256 * 8, iters = 4096;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(d, 16, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst = (double)blocks * 4 /*waves*/ * iters * 32;
    const double lane_ops = inst * 64;
    printf("%-14s %8.3f ms  %7.2f G wave-inst/s  %7.2f TFLOP/s  (cycles/wave-inst/SIMD @2.4GHz: %.2f)\n", name, ms,
           inst / ms / 1e6, lane_ops * flop_per_inst / ms / 1e9, 2.4e9 * ms * 1e-3 * 1024 / inst);
    hipFree(d);
    return ms;
}

int main()
{
    run<0>("v_fma_f32", 2);
    run<1>("v_pk_fma_f32", 4);
    run<2>("v_pk_mul_f32", 2);
    return 0;
}
