/*
 * RaisrHandler.h -- the five-function C ABI that FFmpeg's vf_raisr binds
 * (reference Library/RaisrHandler.h:15-48; call protocol ffmpeg/vf_raisr.c:135-146,286-312,336).
 * Same names, argument order and meaning; implemented in csrc/raisr_api.cpp on top of the
 * raisr_hip_* device ABI (include/raisr_hip.h).
 *
 *   Init     once; loads <modelPath>/{config,filterbin_2_<bits>[_2],Qfactor_*}
 *   SetRes   once, with the first frame's plane descriptors (sizes/steps only)
 *   Process  per frame, synchronous; caller owns every buffer; host planes in, host planes out
 *   SetOpenCLContext(NULL, NULL, platform, device): kept as the device-selection hook --
 *            `deviceIndex` is the HIP device ordinal; call before Init
 *   Deinit   releases device and host resources
 */
#ifndef RAISR_HANDLER_H
#define RAISR_HANDLER_H

#include "RaisrDefaults.h"

#ifdef __cplusplus
extern "C" {
#endif

RNLERRORTYPE RNLHandler_Init(const char *modelPath, float ratio, unsigned int bitDepth,
                             RangeType rangeType, unsigned int threadCount, ASMType asmType,
                             unsigned int passes, unsigned int twoPassMode);

RNLERRORTYPE RNLHandler_SetRes(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                               VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb);

RNLERRORTYPE RNLHandler_Process(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                                VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb,
                                BlendingMode blendingMode);

RNLERRORTYPE RNLHandler_SetOpenCLContext(void *context, void *device_id, int platformIndex,
                                         int deviceIndex);

RNLERRORTYPE RNLHandler_Deinit(void);

#ifdef __cplusplus
}
#endif
#endif /* RAISR_HANDLER_H */
