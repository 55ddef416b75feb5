// masked_load_probe.hip -- does a vector-L1 gather cost less when part of the wave is masked off?
// Mimics the filter stage's coefficient fetch: every 16-lane group reads one 512-B row of an L2-resident 442 KB table
// (2 x dwordx4 per lane), row chosen pseudo-randomly per group and step.  ACTIVE = number of the wave's four groups that
// issue the loads (the others skip them under EXEC).  If the time scales with ACTIVE, a partially hitting LDS cache of
// filter rows pays without compacting hits and misses.
//   hipcc --offload-arch=gfx950 -O3 scripts/masked_load_probe.hip -o /tmp/mlp && /tmp/mlp
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int ACTIVE, int SAMEQUAD>
__global__ __launch_bounds__(256, 3) void probe(const float4* __restrict__ table, float* out, int iters)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    unsigned state = blockIdx.x * 2654435761u + (threadIdx.x >> 4) * 40503u + 12345u;
    float4 acc = {0, 0, 0, 0};
    // SAMEQUAD = 1: the active lanes are lanes with (lane & 3) < ACTIVE instead of whole groups (does a quad have to be empty?)
    const bool on = SAMEQUAD ? ((lane & 3) < ACTIVE) : (g < ACTIVE);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            state = state * 1664525u + 1013904223u;
            const unsigned row = (state >> 10) % 864u;
            if (on) {
                const float4 a = table[row * 32u + l], b = table[row * 32u + 16u + l];
                acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int ACTIVE, int SAMEQUAD>
void run(const float4* table, float* out)
{
    const int blocks = 256 * 12, iters = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<ACTIVE, SAMEQUAD><<<blocks, 256>>>(table, out, 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<ACTIVE, SAMEQUAD><<<blocks, 256>>>(table, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double steps = (double)blocks * 4 * iters * 8;            // wave-steps
    const double clk = ms * 1e-3 * 2.4e9 * 256 / steps;            // CU-cycles per wave-step (4 pixel slots)
    printf("%s active=%d/4: %.3f ms, %.1f CU-clk per wave-step, %.1f per ACTIVE row\n", SAMEQUAD ? "lanes-in-quad" : "groups", ACTIVE, ms, clk, clk / ACTIVE);
}

int main()
{
    float4* table; float* out;
    hipMalloc(&table, 864 * 512); hipMemset(table, 0, 864 * 512);
    hipMalloc(&out, 256 * 12 * 256 * 4);
    run<4, 0>(table, out); run<3, 0>(table, out); run<2, 0>(table, out); run<1, 0>(table, out);
    run<4, 1>(table, out); run<2, 1>(table, out); run<1, 1>(table, out);
    return 0;
}
