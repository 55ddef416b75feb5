// zero_copy_probe.hip -- PCIe ceilings of the streamed host pipeline on this box.
//  (1) one 4K frame (12.4 MB) device -> page-locked host with the copy engine, per hipHostMalloc flag, and with a kernel
//      that stores straight into the mapped host buffer;
//  (2) the ring's steady state: a 12.4 MB download and a 3.1 MB upload in flight at the same time on two streams.
//   hipcc --offload-arch=gfx950 -O3 scripts/zero_copy_probe.hip -o /tmp/zcp && /tmp/zcp
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[i] = src[i];
}

int main()
{
    const size_t out_b = 12441600, in_b = 3110400;
    void *d_out, *d_in; hipMalloc(&d_out, out_b); hipMalloc(&d_in, in_b);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct { const char* name; unsigned flags; } kinds[] = {
        {"Mapped", hipHostMallocMapped}, {"Default", hipHostMallocDefault}, {"Default(again)", hipHostMallocDefault}, {"NumaUser", hipHostMallocNumaUser}, {"NonCoherent", hipHostMallocNonCoherent},
        {"Coherent", hipHostMallocCoherent}, {"Portable", hipHostMallocPortable}, {"WriteCombined", hipHostMallocWriteCombined}};
    for (auto& k : kinds) {
        void *h_out = nullptr, *h_in = nullptr;
        if (hipHostMalloc(&h_out, out_b, k.flags) != hipSuccess || hipHostMalloc(&h_in, in_b, k.flags) != hipSuccess) { printf("%-14s alloc failed\n", k.name); (void)hipGetLastError(); continue; }
        const int reps = 100;
        float ms_d2h, ms_h2d, ms_both, ms_kern = 0;
        hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1);
        hipEventRecord(a, s1); for (int i = 0; i < reps; i++) hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, s1); hipEventRecord(b, s1); hipEventSynchronize(b); hipEventElapsedTime(&ms_d2h, a, b);
        hipEventRecord(a, s1); for (int i = 0; i < reps; i++) hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, s1); hipEventRecord(b, s1); hipEventSynchronize(b); hipEventElapsedTime(&ms_h2d, a, b);
        // steady state of the ring: downloads on s1, uploads on s2, both queues kept full
        hipDeviceSynchronize();
        hipEventRecord(a, s1);
        for (int i = 0; i < reps; i++) { hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, s1); hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, s2); }
        hipStreamSynchronize(s2); hipEventRecord(b, s1); hipEventSynchronize(b); hipEventElapsedTime(&ms_both, a, b);
        void* hdev = nullptr;
        if (hipHostGetDevicePointer(&hdev, h_out, 0) == hipSuccess && hdev) {
            copy_k<<<256, 256, 0, s1>>>((const uint4*)d_out, (uint4*)hdev, out_b / 16); hipStreamSynchronize(s1);
            hipEventRecord(a, s1); for (int i = 0; i < reps; i++) copy_k<<<256, 256, 0, s1>>>((const uint4*)d_out, (uint4*)hdev, out_b / 16); hipEventRecord(b, s1); hipEventSynchronize(b); hipEventElapsedTime(&ms_kern, a, b);
        }
        printf("%-14s D2H %.1f GB/s  H2D %.1f GB/s  kernel-store D2H %.1f GB/s  | D2H+H2D together: %.0f frames/s (%.1f GB/s both ways)\n", k.name,
               out_b * reps / ms_d2h / 1e6, in_b * reps / ms_h2d / 1e6, ms_kern > 0 ? out_b * reps / ms_kern / 1e6 : 0.0,
               reps / ms_both * 1e3, (out_b + in_b) * reps / ms_both / 1e6);
        hipHostFree(h_out); hipHostFree(h_in);
    }
    return 0;
}
