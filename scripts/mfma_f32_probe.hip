// mfma_f32_probe.hip -- v_mfma_f32_16x16x4_f32 on gfx950: (1) operand layout (which lane holds which element), (2) error model of a
// chain of K = 4 steps against float64: max |D - exact| in units of u * (sum of |a b| + |c|), u = 2^-24, on random data with and
// without cancellation.  Decides whether the separable passes of the certified hash stage can run on the matrix cores.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_f32_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// A: [steps][16][4], B: [steps][4][16], D: [16][16]; assumed layout: lane l holds A[l % 16][l / 16], B[l / 16][l % 16], D[4 (l / 16) + v][l % 16]
__global__ void k(const float* A, const float* B, float* D, int steps)
{
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) {
        const float a = A[(s * 16 + (l % 16)) * 4 + l / 16];
        const float b = B[(s * 4 + l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; v++) D[(4 * (l / 16) + v) * 16 + l % 16] = acc[v];
}

int main()
{
    const int steps = 7, trials = 2000;
    float *dA, *dB, *dD;
    hipMalloc(&dA, steps * 64 * 4); hipMalloc(&dB, steps * 64 * 4); hipMalloc(&dD, 256 * 4);
    float hA[7 * 64], hB[7 * 64], hD[256];
    double worst[3] = {0, 0, 0};
    int layout_bad = 0;
    for (int mode = 0; mode < 3; mode++) {       // 0: small integers (layout check, exact), 1: positive random, 2: signed random
        srand(1234 + mode);
        for (int t = 0; t < (mode == 0 ? 4 : trials); t++) {
            for (int i = 0; i < steps * 64; i++) {
                if (mode == 0) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
                else {
                    const float x = (float)rand() / RAND_MAX, y = (float)rand() / RAND_MAX;
                    hA[i] = ldexpf(0.5f + 0.5f * x, rand() % 6 - 3) * ((mode == 2 && (rand() & 1)) ? -1.f : 1.f);
                    hB[i] = ldexpf(0.5f + 0.5f * y, rand() % 6 - 3);
                }
            }
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dD, steps);
            hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                double ex = 0, mag = 0;
                for (int s = 0; s < steps; s++) for (int kk = 0; kk < 4; kk++) {
                    const double p = (double)hA[(s * 16 + i) * 4 + kk] * (double)hB[(s * 4 + kk) * 16 + j];
                    ex += p; mag += fabs(p);
                }
                const double err = fabs((double)hD[i * 16 + j] - ex);
                if (mode == 0) { if (err != 0) layout_bad++; }
                else { const double r = err / (mag * ldexp(1.0, -24)); if (r > worst[mode]) worst[mode] = r; }
            }
        }
    }
    printf("layout (integers, exact): %s\n", layout_bad ? "MISMATCH" : "as assumed");
    printf("7 chained K=4 steps: max |D - exact| / (u sum|ab|): positive data %.3f, signed data %.3f  (28 sequential RNE FMAs would allow <= 28)\n", worst[1], worst[2]);
    // single step, does it round once?  c + 4 products with exactly representable sum + a half-ulp tie
    return 0;
}
