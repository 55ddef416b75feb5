// trans_rate_probe.hip -- issue cost of v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 on gfx950 against v_fma_f32, alone and mixed 1 : 7 with
// FMAs (does a transcendental hide behind plain vector ALU work of the same wave?).  Decides what a reciprocal in approx_hash costs.
//   hipcc --offload-arch=gfx950 -O3 scripts/trans_rate_probe.hip -o /tmp/trans_probe && /tmp/trans_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float s)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.25f);
                if (MODE == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
                if (MODE == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
                if (MODE == 3) a[i] = __builtin_amdgcn_rsqf(a[i]);
                if (MODE == 4) a[i] = (i == 0) ? __builtin_amdgcn_rcpf(a[i]) : __builtin_fmaf(a[i], s, 0.25f);      // 1 rcp : 7 fma
                if (MODE == 5) a[i] = (i % 4 == 0) ? __builtin_amdgcn_rcpf(a[i]) : __builtin_fmaf(a[i], s, 0.25f);  // 1 rcp : 3 fma
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
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(d, 16, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst = (double)blocks * 4 /*waves*/ * iters * 32;
    printf("%-22s %d WG/CU %8.3f ms  %7.2f G wave-inst/s  (cycles per wave-inst and SIMD @2.4 GHz: %.2f)\n", name, blocks_per_cu, ms,
           inst / ms / 1e6, 2.4e9 * ms * 1e-3 * 1024 / inst);
    hipFree(d);
}

int main()
{
    for (int wg = 4; wg <= 8; wg += 4) {
        run<0>("v_fma_f32", wg);
        run<1>("v_rcp_f32", wg);
        run<2>("v_sqrt_f32", wg);
        run<3>("v_rsq_f32", wg);
        run<4>("1 rcp : 7 fma", wg);
        run<5>("1 rcp : 3 fma", wg);
    }
    return 0;
}
