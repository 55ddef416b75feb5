// l1_width_probe.hip -- what does the vector L1 deliver per clock for the filter stage's coefficient gather, by load width?
// Every 16-lane group reads one 512-B row of a table (NROWS rows: 16 = 8 KB, vector-L1 resident; 864 = the 442 KB bank, L2
// resident), row chosen pseudo-randomly per group and step, with 8 x dword (the production kernel's pattern: lane l reads
// floats l, 16 + l, ...), 4 x dwordx2 or 2 x dwordx4 per lane.  Same bytes per step (2048 B per wave); if the wide forms
// are faster, the L1's limit is instructions (address coalescing), not bytes.
//   hipcc --offload-arch=gfx950 -O3 scripts/l1_width_probe.hip -o /tmp/l1p && /tmp/l1p
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int WIDTH, int NROWS, int WGS>
__global__ __launch_bounds__(256, WGS) void probe(const float* __restrict__ table, float* out, int iters)
{
    const int lane = threadIdx.x & 63, l = lane & 15;
    unsigned state = blockIdx.x * 2654435761u + (threadIdx.x >> 4) * 40503u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            state = state * 1664525u + 1013904223u;
            const float* row = table + ((state >> 10) % (unsigned)NROWS) * 128u;
            if (WIDTH == 4) {
#pragma unroll
                for (int ch = 0; ch < 8; ch++) acc += row[16 * ch + l];
            } else if (WIDTH == 8) {
#pragma unroll
                for (int ch = 0; ch < 4; ch++) { const float2 v = reinterpret_cast<const float2*>(row)[16 * ch + l]; acc += v.x + v.y; }
            } else {
#pragma unroll
                for (int ch = 0; ch < 2; ch++) { const float4 v = reinterpret_cast<const float4*>(row)[16 * ch + l]; acc += (v.x + v.y) + (v.z + v.w); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH, int NROWS, int WGS>
void run(const float* table, float* out)
{
    const int blocks = 256 * WGS, iters = 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<WIDTH, NROWS, WGS><<<blocks, 256>>>(table, out, 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<WIDTH, NROWS, WGS><<<blocks, 256>>>(table, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 4 * iters * 4 * 2048.0;   // per wave-step: 4 rows x 512 B
    printf("width %2d B, %3d rows, %d waves/CU: %.3f ms, %.1f B/clk/CU at 2.4 GHz, %.2f TB/s\n", WIDTH, NROWS, 4 * WGS, ms,
           bytes / (ms * 1e-3 * 2.4e9 * 256), bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    float* table; float* out;
    hipMalloc(&table, 864 * 512); hipMemset(table, 0, 864 * 512);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<4, 16, 4>(table, out); run<8, 16, 4>(table, out); run<16, 16, 4>(table, out);
    run<4, 16, 8>(table, out); run<8, 16, 8>(table, out); run<16, 16, 8>(table, out);
    run<4, 864, 4>(table, out); run<8, 864, 4>(table, out); run<16, 864, 4>(table, out);
    run<4, 864, 8>(table, out); run<8, 864, 8>(table, out); run<16, 864, 8>(table, out);
    run<4, 64, 4>(table, out); run<16, 64, 4>(table, out); run<4, 216, 4>(table, out); run<16, 216, 4>(table, out);
    return 0;
}
