// mfma_valu_overlap_probe.hip -- does v_mfma_f32_16x16x4_f32 (fp32 in) run beside another wave's v_fma_f32 on the same SIMD, as the
// bf16 matrix instructions do?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (one per SIMD)
// issue FMAs; each half alone, then both.  Concurrent pipes: t(both) ~ max; a shared pipe: t(both) ~ sum.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND, int PRIO = 0>      // 0: f32 16x16x4, 1: bf16 16x16x32;  PRIO 1: the FMA waves run at s_setprio 1, 2: the MFMA waves do
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_fma)
{
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (PRIO == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && w < 4) __builtin_amdgcn_s_setprio(1);
    if (w < 4) {
        f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        bf8 xb, yb;
        for (int i = 0; i < 8; i++) { xb[i] = (__bf16)x; yb[i] = (__bf16)y; }
        for (int i = 0; i < n_mfma; i++) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        float a[8];
        for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < n_fma; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], 0.999f, 0.25f);
        for (int i = 0; i < 8; i++) r += a[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND, int PRIO = 0>
float run(int n_mfma, int n_fma)
{
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<KIND, PRIO><<<256, 512>>>(d, 8, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    probe<KIND, PRIO><<<256, 512>>>(d, n_mfma, n_fma);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms;
}

int main()
{
    const int NM = 1 << 16;                  // x 4 MFMAs
    for (int kind = 0; kind < 2; kind++) {
        const float tm = kind == 0 ? run<0>(NM, 0) : run<1>(NM, 0);
        // FMA count that takes about as long as the MFMAs alone
        const float tf1 = run<0>(0, 1 << 16);
        const int NF = (int)((double)(1 << 16) * tm / tf1);
        const float tf = run<0>(0, NF);
        const float tb = kind == 0 ? run<0>(NM, NF) : run<1>(NM, NF);
        printf("%s: MFMA waves alone %.3f ms (%.1f cycles per MFMA and SIMD @2.4 GHz), FMA waves alone %.3f ms, both %.3f ms  -> both / max = %.2f, both / sum = %.2f\n",
               kind == 0 ? "v_mfma_f32_16x16x4_f32  " : "v_mfma_f32_16x16x32_bf16", tm, tm * 1e-3 * 2.4e9 / (4.0 * NM), tf, tb,
               tb / (tm > tf ? tm : tf), tb / (tm + tf));
        const float tp1 = kind == 0 ? run<0, 1>(NM, NF) : run<1, 1>(NM, NF);
        const float tp2 = kind == 0 ? run<0, 2>(NM, NF) : run<1, 2>(NM, NF);
        printf("    both, FMA waves at s_setprio 1: %.3f ms;  both, MFMA waves at s_setprio 1: %.3f ms\n", tp1, tp2);
    }
    return 0;
}
