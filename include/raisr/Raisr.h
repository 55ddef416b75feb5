/*
 * Raisr.h -- C++ flavour of the raisr API (reference Library/Raisr.h:14-33): identical
 * signatures and default arguments; the RNLHandler_* C functions forward to these.
 */
#ifndef RAISR_H
#define RAISR_H

#include <string>
#include <vector>
#include "RaisrDefaults.h"
#include "RaisrVersion.h"

RNLERRORTYPE RNLInit(std::string &modelPath, float ratio, unsigned int bitDepth = 8,
                     RangeType rangeType = VideoRange, unsigned int threadCount = 20,
                     ASMType asmType = AVX512, unsigned int passes = 1,
                     unsigned int twoPassMode = 1);

RNLERRORTYPE RNLSetRes(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                       VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb);

RNLERRORTYPE RNLProcess(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                        VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb,
                        BlendingMode blendingMode = CountOfBitsChanged);

RNLERRORTYPE RNLSetOpenCLContext(void *context, void *device_id, int platformIndex,
                                 int deviceIndex);

RNLERRORTYPE RNLDeinit();

#endif /* RAISR_H */
