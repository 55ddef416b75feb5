/*
 * RaisrDefaults.h -- public types of the raisr C/C++ API, MI355X (HIP) implementation.
 *
 * Binary-compatible with the reference's Library/RaisrDefaults.h:13-57 so that existing callers
 * (ffmpeg/vf_raisr.c) compile and link unchanged: same struct field order, same enumerator
 * values.  One enumerator is appended (never renumbered): ASMType::HIP = 6 selects the gfx950
 * backend explicitly; every other accepted ASMType value also runs on the GPU and selects which
 * x86 code path's results are reproduced bit-for-bit (see DESIGN.md "asm mapping").
 */
#ifndef RAISR_DEFAULTS_H
#define RAISR_DEFAULTS_H

#define defaultPatchSize (11)
static const unsigned int defaultPatchAreaSize = defaultPatchSize * defaultPatchSize;

/* One image plane.  `step` is the byte distance between rows and may exceed width*bytes. */
typedef struct VideoDataType {
    unsigned char *pData;
    unsigned int   width;
    unsigned int   height;
    unsigned int   step;
    unsigned int   bitShift;   /* unused by this backend (as on the reference's CPU path) */
} VideoDataType;

typedef enum RNLERRORTYPE {
    RNLErrorNone                  = 0,
    RNLErrorInsufficientResources = (int)0x80001000,
    RNLErrorUndefined             = (int)0x80001001,
    RNLErrorBadParameter          = (int)0x80001002,
    RNLErrorMax                   = (int)0x7FFFFFFF
} RNLERRORTYPE;

typedef enum BlendingMode {
    Randomness         = 1,
    CountOfBitsChanged = 2
} BlendingMode;

typedef enum ASMType {
    AVX2           = 1,   /* GPU reproduces the AVX2 path's output            */
    AVX512         = 2,   /* GPU reproduces the AVX-512 fp32 path's output    */
    OpenCL         = 3,   /* rejected: no OpenCL in this build                */
    OpenCLExternal = 4,   /* rejected                                         */
    AVX512_FP16    = 5,   /* GPU reproduces the AVX512-FP16 path (8-bit)      */
    HIP            = 6    /* appended: MI355X backend, AVX-512 fp32 numerics  */
} ASMType;

typedef enum MachineVendorType {
    INTEL              = 1,
    AMD                = 2,
    VENDOR_UNSUPPORTED = 3
} MachineVendorType;

typedef enum RangeType {
    VideoRange = 1,
    FullRange  = 2
} RangeType;

#endif /* RAISR_DEFAULTS_H */
