/*
 * Raisr.h -- C++ flavour of the raisr API.  The five entry points below replace the ones the
 * reference declares in Library/Raisr.h:14-33 (same types, order and default arguments, so
 * existing C++ callers recompile unchanged); the RNLHandler_* C functions forward to them.
 * Implementation: video-super-resolution-library_amd/csrc/raisr_api.cpp (MI355X / HIP backend).
 */
#ifndef RAISR_H
#define RAISR_H

#include <string>
#include <vector>
#include "RaisrDefaults.h"
#include "RaisrVersion.h"

/*
 * Load the trained filter bank, quantisation thresholds and (2-pass) the second bank from
 * `filterFolder`, pick the numerics (`numerics`: AVX2 / AVX512 / AVX512_FP16 select which CPU
 * path's results are reproduced bit-exactly on the GPU; HIP == AVX512 numerics) and create the
 * device context.  `workers` is accepted for compatibility and ignored (one frame = one launch
 * sequence).  Replaces Library/Raisr.cpp:1409 RNLInit.
 */
RNLERRORTYPE RNLInit(std::string &filterFolder, float upscaleRatio,
                     unsigned int sampleBits = 8,
                     RangeType range = VideoRange,
                     unsigned int workers = 20,
                     ASMType numerics = AVX512,
                     unsigned int numPasses = 1,
                     unsigned int passMode = 1);

/*
 * Declare the plane geometry of the stream (input and output Y/Cr/Cb descriptors; only sizes and
 * steps are read) and allocate the HBM planes.  Replaces Library/Raisr.cpp:1681 RNLSetRes.
 */
RNLERRORTYPE RNLSetRes(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                       VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb);

/*
 * Upscale one frame: Y through the RAISR path, Cr/Cb through the cheap upscale.  Synchronous;
 * no pData pointer is retained.  Replaces Library/Raisr.cpp:1294 RNLProcess.
 */
RNLERRORTYPE RNLProcess(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                        VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb,
                        BlendingMode blend = CountOfBitsChanged);

/*
 * Device selection hook (reference: OpenCL context hand-over, Library/Raisr.cpp:1399).  Here
 * `deviceOrdinal` is the HIP device ordinal; `clContext` / `clDevice` / `platformOrdinal` are
 * ignored.  Must be called before RNLInit.
 */
RNLERRORTYPE RNLSetOpenCLContext(void *clContext, void *clDevice, int platformOrdinal,
                                 int deviceOrdinal);

/* Release the device context and all HBM planes.  Replaces Library/Raisr.cpp:1842 RNLDeinit. */
RNLERRORTYPE RNLDeinit();

/* Asynchronous frames (extension, see RaisrHandler.h): up to `depth` frames in flight. */
RNLERRORTYPE RNLSetAsyncDepth(unsigned int depth);
int RNLAsyncCapacity();
RNLERRORTYPE RNLSetDeviceList(const char *devices);     /* "0,1,2" | "all" | "": GPUs of the asynchronous ring (RaisrHandler.h) */
RNLERRORTYPE RNLSubmit(VideoDataType *srcY, VideoDataType *srcCr, VideoDataType *srcCb,
                       VideoDataType *dstY, VideoDataType *dstCr, VideoDataType *dstCb,
                       BlendingMode blend = CountOfBitsChanged);
RNLERRORTYPE RNLCollect();
int RNLFramesInFlight();

/* Page-locked frame memory (extension, see RaisrHandler.h). */
void *RNLHostAlloc(size_t bytes);
void RNLHostFree(void *p);

#endif /* RAISR_H */
