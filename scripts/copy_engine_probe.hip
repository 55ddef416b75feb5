// copy_engine_probe.hip -- when does hipMemcpyAsync(device -> page-locked host) go to the DMA engine and when to a copy kernel?
// Run under: rocprofv3 --kernel-trace --memory-copy-trace --stats ... and count MEMORY_COPY_DEVICE_TO_HOST vs __amd_rocclr_copyBuffer.
// usage: copy_engine_probe <variant>   0: kernel -> D2H, same stream, deep queue
//                                      1: H2D on stream U, event; stream C waits event -> kernel -> D2H
//                                      2: H2D, kernel, D2H all on one stream
//                                      3: like 1, but the host waits for the previous frame's download before submitting the next
//                                      4: like 1 with THREE compute streams taken in turn (the ring), host waits for frame f-3
//                                      5: like 4 without the upload stream (uploads on the compute streams)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void spin(float* p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main(int argc, char** argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t out_b = 12441600, in_b = 3110400;
    void *d_out, *d_in, *h_out, *h_in; float* d_w;
    hipMalloc(&d_out, out_b); hipMalloc(&d_in, in_b); hipMalloc(&d_w, 1024 * 256 * 4);
    hipHostMalloc(&h_out, out_b, hipHostMallocDefault); hipHostMalloc(&h_in, in_b, hipHostMallocDefault);
    hipStream_t U, C; hipStreamCreateWithFlags(&U, hipStreamNonBlocking); hipStreamCreateWithFlags(&C, hipStreamNonBlocking);
    hipEvent_t ev, done; hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventCreateWithFlags(&done, hipEventDisableTiming);
    hipStream_t C3[3]; hipEvent_t ev3[3], done3[3];
    void *d_out3[3], *h_out3[3];
    for (int i = 0; i < 3; i++) { hipStreamCreateWithFlags(&C3[i], hipStreamNonBlocking); hipEventCreateWithFlags(&ev3[i], hipEventDisableTiming); hipEventCreateWithFlags(&done3[i], hipEventDisableTiming);
                                  hipMalloc(&d_out3[i], out_b); hipHostMalloc(&h_out3[i], out_b, hipHostMallocDefault); }
    if (variant >= 4) {
        for (int f = 0; f < 60; f++) {
            const int i = f % 3;
            if (f >= 3) hipEventSynchronize(done3[i]);
            if (variant == 4) { hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, U); hipEventRecord(ev3[i], U); hipStreamWaitEvent(C3[i], ev3[i], 0); }
            else hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, C3[i]);
            spin<<<1024, 256, 0, C3[i]>>>(d_w, 20000);
            hipMemcpyAsync(h_out3[i], d_out3[i], out_b, hipMemcpyDeviceToHost, C3[i]);
            hipEventRecord(done3[i], C3[i]);
        }
        hipDeviceSynchronize();
        printf("variant %d done\n", variant);
        return 0;
    }
    for (int f = 0; f < 40; f++) {
        if (variant == 1 || variant == 3) { hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, U); hipEventRecord(ev, U); hipStreamWaitEvent(C, ev, 0); }
        if (variant == 2) hipMemcpyAsync(d_in, h_in, in_b, hipMemcpyHostToDevice, C);
        spin<<<1024, 256, 0, C>>>(d_w, 20000);
        hipMemcpyAsync(h_out, d_out, out_b, hipMemcpyDeviceToHost, C);
        if (variant == 3) { hipEventRecord(done, C); hipEventSynchronize(done); }
    }
    hipDeviceSynchronize();
    printf("variant %d done\n", variant);
    return 0;
}
