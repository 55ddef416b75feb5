// mfma_assist_probe.hip -- can the MFMA pipe take the non-packed FMA chain of the structure tensor off the VALU?
// v_mfma_f32_4x4x1_16b_f32 computes, per lane l, acc[i] += A[4*(l/4)+i] * B[l] for i = 0..3; component i == l%4 is the
// lane's own a[l]*b[l] (the other three are cross terms that are thrown away), and an f32 MFMA is bit-identical to fmaf.
// MODE 0: per tap  v_pk_mul + v_pk_fma + v_fma      (what the hash stage issues today)
// MODE 1: per tap  v_pk_mul + v_pk_fma + mfma 4x4x1 (B chain on the matrix pipe)
// Build & run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/mfma_assist_probe.hip -o /tmp/mp && /tmp/mp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(float* out, int iters, float s)
{
    f2 g[4], AD[4]; float B[4]; f4 BM[4];
    for (int j = 0; j < 4; j++) {
        g[j] = (f2){threadIdx.x * 0.001f + j, threadIdx.x * 0.002f - j};
        AD[j] = (f2){0.f, 0.f}; B[j] = 0.f; BM[j] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float wv = s + i * 1e-3f;
            const f2 w2 = {wv, wv};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f2 pq = g[j] * w2;
                AD[j] = __builtin_elementwise_fma(pq, g[j], AD[j]);
                if (MODE == 0) B[j] = __builtin_fmaf(pq.x, g[j].y, B[j]);
                else BM[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(pq.x, g[j].y, BM[j], 0, 0, 0);
                g[j].x += 1e-7f;       // keep the compiler from hoisting
            }
        }
    }
    float r = 0;
    for (int j = 0; j < 4; j++) r += AD[j].x + AD[j].y + B[j] + BM[j][threadIdx.x & 3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name)
{
    const int blocks = 256 * 8, iters = 512;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(d, 8, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double taps = (double)blocks * 4 * iters * 44;      // wave-level (tap, pixel) pairs
    printf("%-34s %8.3f ms   %.2f G wave-taps/s\n", name, ms, taps / ms / 1e6);
    hipFree(d);
}

int main()
{
    run<0>("VALU only (pk_mul+pk_fma+fma)");
    run<1>("VALU + MFMA (pk_mul+pk_fma | mfma)");
    // correctness of the diagonal: acc[l%4] == fmaf chain
    return 0;
}
