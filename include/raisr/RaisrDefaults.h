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

/* ---- status codes returned by every RNL* / RNLHandler_* call --------------------------------- */
typedef enum RNLERRORTYPE {
    RNLErrorNone                  = 0,
    RNLErrorInsufficientResources = (int)0x80001000,   /* host or HBM allocation failed          */
    RNLErrorUndefined             = (int)0x80001001,   /* HIP runtime / launch failure           */
    RNLErrorBadParameter          = (int)0x80001002,   /* rejected argument or missing model file */
    RNLErrorMax                   = (int)0x7FFFFFFF
} RNLERRORTYPE;

/* ---- which numerics the GPU reproduces (the reference's "asm" option) ------------------------ */
typedef enum ASMType {
    AVX2           = 1,   /* AVX2 path's output (RCPPS/RSQRTPS-based hashing)           */
    AVX512         = 2,   /* AVX-512 fp32 path's output (VRCP14/VRSQRT14-based hashing) */
    OpenCL         = 3,   /* rejected: no OpenCL in this build                          */
    OpenCLExternal = 4,   /* rejected                                                   */
    AVX512_FP16    = 5,   /* AVX512-FP16 path's output (binary16 pipeline, 8-bit only)  */
    HIP            = 6,   /* appended: MI355X backend, AVX-512 fp32 numerics            */
    HIPExternal    = 7    /* appended: as HIP, but every VideoDataType::pData handed to SetRes / Process is a HIP
                           * DEVICE pointer (step = device pitch in bytes): frames that are already in HBM (hardware
                           * decode, a previous GPU filter) are processed in place -- the counterpart of the reference's
                           * OpenCLExternal / vf_raisr_opencl.c zero-copy path (ffmpeg/vf_raisr_opencl.c:50-150)      */
} ASMType;

/* ---- sample range: VideoRange clamps to [16,235]<<(bits-8), FullRange to [0,2^bits-1] -------- */
typedef enum RangeType {
    VideoRange = 1,
    FullRange  = 2
} RangeType;

/* ---- how the filtered and the cheap-upscaled planes are merged ------------------------------- */
typedef enum BlendingMode {
    Randomness         = 1,   /* 3x3 census "randomness" weight between the two planes */
    CountOfBitsChanged = 2    /* census bits changed -> per-pixel weight (default)      */
} BlendingMode;

/* Kept for source compatibility; the HIP backend does not look at the host CPU vendor. */
typedef enum MachineVendorType {
    INTEL              = 1,
    AMD                = 2,
    VENDOR_UNSUPPORTED = 3
} MachineVendorType;

/* ---- one image plane; `step` = bytes between rows (may exceed width * bytes per sample) ------ */
typedef struct VideoDataType {
    unsigned char *pData;      /* host pointer, caller-owned (a HIP device pointer with asm = HIPExternal) */
    unsigned int   width;      /* samples per row                                                */
    unsigned int   height;     /* rows                                                           */
    unsigned int   step;
    unsigned int   bitShift;   /* low 8 bits: samples are MSB-aligned by this many bits -- honoured for device frames (asm = HIPExternal), as by the reference's OpenCL path; ignored for host planes, as by its CPU path; top bit: RAISR_HIP_INTERLEAVED2 */
} VideoDataType;

/* Extension for device frames (asm = HIPExternal) whose chroma is ONE plane of interleaved (U, V) pairs (NV12 / P010, what
 * hardware decoders produce): set this bit in the bitShift of BOTH chroma descriptors, point them at the plane's first U and
 * first V sample (one sample apart), give `width` in samples of one channel and `step` as the plane's pitch in bytes. */
#define RAISR_HIP_INTERLEAVED2 0x80000000u

/* ---- filter geometry: 11x11 patch, 121 taps --------------------------------------------------- */
#define defaultPatchSize (11)
static const unsigned int defaultPatchAreaSize = defaultPatchSize * defaultPatchSize;

#endif /* RAISR_DEFAULTS_H */
