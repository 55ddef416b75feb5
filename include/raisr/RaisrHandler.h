/*
 * RaisrHandler.h -- the five-function C ABI that FFmpeg's vf_raisr binds
 * (reference Library/RaisrHandler.h:15-48; call protocol ffmpeg/vf_raisr.c:135-146,286-312,336).
 * Same names, argument order and meaning; implemented in csrc/raisr_api.cpp on top of the
 * raisr_hip_* device ABI (include/raisr_hip.h).
 *
 *   Init     once; loads <filterFolder>/{config,filterbin_2_<bits>[_2],Qfactor_*}
 *   SetRes   once, with the first frame's plane descriptors (sizes/steps only)
 *   Process  per frame, synchronous; caller owns every buffer; host planes in, host planes out
 *   SetOpenCLContext(NULL, NULL, platform, device): kept as the device-selection hook --
 *            `deviceOrdinal` is the HIP device ordinal; call before Init.
 *            With asm = HIPExternal (device-pointer planes) `clContext` may carry the caller's hipStream_t: Process then
 *            enqueues on that stream and returns without waiting (stream-ordered, like the reference's external OpenCL
 *            queue); with clContext == NULL Process waits for the frame before it returns.
 *   Deinit   releases device and host resources
 */
#ifndef RAISR_HANDLER_H
#define RAISR_HANDLER_H

#include <stddef.h>
#include "RaisrDefaults.h"

#ifdef __cplusplus
extern "C" {
#endif

/* -> RNLInit (Raisr.h); the C string is copied. */
RNLERRORTYPE RNLHandler_Init(const char *filterFolder, float upscaleRatio, unsigned int sampleBits,
                             RangeType range, unsigned int workers, ASMType numerics,
                             unsigned int numPasses, unsigned int passMode);

/* -> RNLSetRes */
RNLERRORTYPE RNLHandler_SetRes(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                               VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb);

/* -> RNLProcess */
RNLERRORTYPE RNLHandler_Process(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                                VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb,
                                BlendingMode blend);

/* -> RNLSetOpenCLContext */
RNLERRORTYPE RNLHandler_SetOpenCLContext(void *clContext, void *clDevice, int platformOrdinal,
                                         int deviceOrdinal);

/* -> RNLDeinit */
RNLERRORTYPE RNLHandler_Deinit(void);

/*
 * Asynchronous frames -- an EXTENSION of the reference's surface (its Process is synchronous, Raisr.cpp:1294-1397): what a
 * filter with frame queuing (ffmpeg/vf_raisr_hip.diff, option async=N) uses to keep N frames in flight so that uploads, kernels
 * and downloads of neighbouring frames overlap.  Same validation, same bits as Process.
 *   SetAsyncDepth(N)   after Init, before the first Submit; N = 0 releases the ring; N <= 4 (RNLErrorBadParameter above)
 *   Submit(...)        enqueue one frame and return; RNLErrorInsufficientResources = N frames already in flight (Collect first).
 *                      Every plane must stay valid and untouched until the frame's Collect returns.
 *   Collect()          wait for the OLDEST submitted frame; its output planes are complete on return
 *   FramesInFlight()   submitted and not yet collected
 * Deinit waits for the frames in flight and drops them; SetRes refuses (RNLErrorBadParameter) while frames are in flight:
 * collect them first.  Not available with asm = HIPExternal (device frames are stream-ordered).
 *   SetDeviceList("0,1,2,3" | "all" | "")   SEVERAL GPUs behind one handler: the asynchronous ring gets `depth` lanes on every
 *                      listed device, frame i runs on device i mod n, Collect keeps submission order -- one FFmpeg process
 *                      feeds the whole node (the reference reaches its socket numbers with N processes, docs/performance.md:8-13).
 *                      "" = the single device of SetOpenCLContext / RAISR_HIP_DEVICE again.  Without a call the environment
 *                      variable RAISR_HIP_DEVICES (same syntax) is honoured.  Call after Init with nothing in flight;
 *                      RNLErrorBadParameter for a malformed list or a device the runtime does not have.  Process (synchronous)
 *                      keeps running on the first device.
 */
RNLERRORTYPE RNLHandler_SetAsyncDepth(unsigned int depth);
RNLERRORTYPE RNLHandler_SetDeviceList(const char *devices);
int RNLHandler_AsyncCapacity(void);        /* frames Submit accepts before a Collect is due: depth x the GPUs of the device list (0: no ring asked for) */
RNLERRORTYPE RNLHandler_Submit(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                               VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb,
                               BlendingMode blend);
RNLERRORTYPE RNLHandler_Collect(void);
int RNLHandler_FramesInFlight(void);

/*
 * Page-locked frame memory -- EXTENSION.  A copy from or to pageable memory is staged by the HIP runtime on the calling thread;
 * on page-locked memory it is a DMA that runs next to the kernels.  A host that takes its frame buffers from here (FFmpeg: the
 * buffer pools ffmpeg/vf_raisr_hip.diff installs) gets that without the library ever touching memory it does not own:
 * Process recognises page-locked planes and then returns finished rows while later rows are computed; Submit / Collect overlap
 * whole frames.  Ordinary (malloc'ed) planes keep working everywhere, at the staged rate.  HostFree may be called at any time,
 * also after Deinit; NULL is ignored.
 */
void *RNLHandler_HostAlloc(size_t bytes);
void RNLHandler_HostFree(void *p);

#ifdef __cplusplus
}
#endif
#endif /* RAISR_HANDLER_H */
