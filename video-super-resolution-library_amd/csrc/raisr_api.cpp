// raisr_api.cpp -- host side of the drop-in library: the reference's RNL* C++ API and RNLHandler_*
// C ABI (reference Library/Raisr.h:14-33, Library/RaisrHandler.h:15-48) implemented over the thin
// raisr_hip_* device ABI (include/raisr_hip.h).  Plain C++17; no HIP types in this file.
//
// Behaviour mirrored from the reference (same return codes and the same stdout messages, because
// the reference's validation suite greps them -- test/validation_suite/run_tests_avxout.sh:109-178):
//   parameter validation + banner              RNLInit            Library/Raisr.cpp:1409-1539
//   `config` parsing                           RNLInit/RNLStoi    Library/Raisr.cpp:213-244,1531-1578
//   filterbin / Qfactor loading + validation   ReadTrainedData    Library/Raisr.cpp:246-433
//                                              VerifyTrainedData  Library/Raisr.cpp:187-211
//   first-frame geometry                       RNLSetRes          Library/Raisr.cpp:1681-1829
//   per-frame null checks + Y/U/V dispatch     RNLProcess         Library/Raisr.cpp:1294-1397
// State is one process-global instance, as in the reference (Library/Raisr_globals.h:140-203).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/raisr/Raisr.h"
#include "../../include/raisr/RaisrHandler.h"
#include "../../include/raisr_hip.h"

namespace {

struct PassModel {
    std::vector<float> bank;      // [hashkeys][pixelTypes][121]
    unsigned hashkeys = 0, pixelTypes = 0, rows = 0;
    std::vector<double> qstr, qcoh;   // std::stod values; the device layer derives (float) and (_Float16)
};

struct State {
    bool inited = false, resSet = false;
    int device = 0;
    float ratio = 2.0f;
    unsigned bitDepth = 8;
    int lo = 16, hi = 235;
    unsigned passes = 1, twoPassMode = 1;
    int hashVariant = RAISR_HIP_HASH_AVX512;
    unsigned qAngle = 0, qStrength = 0, qCoherence = 0, patchSize = 0;
    PassModel model[2];
    raisr_hip_ctx *ctx = nullptr;
} G;

// ---- config / trained-data parsing -------------------------------------------------------------

RNLERRORTYPE parseUnsigned(unsigned *value, const std::string &token, const std::string &configPath)
{
    // std::stoi semantics of RNLStoi (Raisr.cpp:213-244): leading integer prefix, negative rejected
    try {
        int v = std::stoi(token);
        if (v < 0) throw std::out_of_range("negative");
        *value = (unsigned)v;
        return RNLErrorNone;
    } catch (const std::exception &) {
        std::cout << "[RAISR ERROR] configFile corrupted: " << configPath << std::endl;
        return RNLErrorBadParameter;
    }
}

// VerifyTrainedData (Raisr.cpp:187-211): token may contain only "-.0123456789", at most one dot, not
// leading, and no dot before a minus sign.
bool tokenLooksNumeric(const std::string &tok)
{
    for (char ch : tok)
        if (ch < '-' || ch > '9' || ch == '/') return false;
    const size_t firstDot = tok.find_first_of('.'), lastDot = tok.find_last_of('.');
    if (firstDot != lastDot || firstDot == 0) return false;
    const size_t firstMinus = tok.find_first_of('-');
    if (firstMinus < 0xFFFF && firstDot < firstMinus) return false;
    return true;
}

RNLERRORTYPE readThresholds(const std::string &path, const char *kind, unsigned expected, std::vector<double> &out)
{
    std::ifstream f(path);
    if (!f.is_open()) {
        std::cout << "[RAISR ERROR] Unable to load model: " << path << std::endl;
        return RNLErrorBadParameter;
    }
    out.clear();
    std::string tok;
    try {
        while (f >> tok) {
            if (!tokenLooksNumeric(tok)) {
                std::cout << "[RAISR ERROR] " << kind << " corrupted: " << path << std::endl;
                return RNLErrorBadParameter;
            }
            out.push_back(std::stod(tok));
        }
    } catch (const std::exception &) {
        std::cout << "[RAISR ERROR] " << kind << " corrupted: " << path << std::endl;
        return RNLErrorBadParameter;
    }
    if (out.size() != expected) {
        std::cout << "[RAISR ERROR] " << kind << " corrupted: " << path << std::endl;
        return RNLErrorBadParameter;
    }
    return RNLErrorNone;
}

RNLERRORTYPE readTrainedData(std::string hashtablePath, std::string strPath, std::string cohPath, int pass, PassModel &M)
{
    if (pass == 2) { hashtablePath += "_2"; strPath += "_2"; cohPath += "_2"; }
    std::ifstream f(hashtablePath, std::ifstream::binary);
    if (!f.is_open()) {
        std::cout << "[RAISR ERROR] Unable to load model: " << hashtablePath << std::endl;
        return RNLErrorBadParameter;
    }
    f.seekg(0, f.end);
    const long fileSize = (long)f.tellg();
    f.seekg(0, f.beg);
    char tag[5] = {0, 0, 0, 0, 0};
    f.read(tag, 4);
    const std::string dataType(tag);
    if (dataType != "fp32" && dataType != "fp16") {
        std::cout << "[RAISR ERROR] hashtable corrupted: " << hashtablePath << std::endl;
        return RNLErrorBadParameter;
    }
    const unsigned weightBytes = dataType == "fp16" ? 2 : 4;
    uint32_t hdr[3] = {0, 0, 0};
    f.read(reinterpret_cast<char *>(hdr), sizeof hdr);
    M.hashkeys = hdr[0]; M.pixelTypes = hdr[1]; M.rows = hdr[2];
    const long headSize = 4 + 3 * 4;
    if ((fileSize - headSize) != (long)((uint64_t)M.hashkeys * M.pixelTypes * M.rows * weightBytes)) {
        std::cout << "[RAISR ERROR] hashtable corrupted: " << hashtablePath << std::endl;
        return RNLErrorBadParameter;
    }
    if (M.hashkeys != G.qAngle * G.qStrength * G.qCoherence) {
        std::cout << "[RAISR ERROR] HashTable format is not compatible in number of hash keys!\n";
        std::cout << M.hashkeys << std::endl;
        return RNLErrorBadParameter;
    }
    if (M.pixelTypes != (unsigned)((int)G.ratio * (int)G.ratio)) {
        std::cout << "[RAISR ERROR] HashTable format is not compatible in number of pixel types!\n";
        return RNLErrorBadParameter;
    }
    if (G.patchSize % 2 == 0 || M.rows != G.patchSize * G.patchSize) {
        std::cout << "[RAISR ERROR] HashTable format is not compatible in patch size!\n";
        return RNLErrorBadParameter;
    }
    if (weightBytes != 4) {
        // the reference's float loader rejects fp16 payloads as well (Raisr.cpp:353-356)
        std::cout << "[RAISR ERROR] hashtable corrupted: " << hashtablePath << std::endl;
        return RNLErrorBadParameter;
    }
    M.bank.resize((size_t)M.hashkeys * M.pixelTypes * M.rows);
    f.read(reinterpret_cast<char *>(M.bank.data()), (std::streamsize)(M.bank.size() * sizeof(float)));
    f.close();
    if (RNLErrorNone != readThresholds(strPath, "StrFile", G.qStrength - 1, M.qstr)) return RNLErrorBadParameter;
    if (RNLErrorNone != readThresholds(cohPath, "CohFile", G.qCoherence - 1, M.qcoh)) return RNLErrorBadParameter;
    return RNLErrorNone;
}

void dropContext()
{
    if (G.ctx) { raisr_hip_destroy(G.ctx); G.ctx = nullptr; }
    G.resSet = false;
}

}  // namespace

// ---- C++ API -----------------------------------------------------------------------------------

RNLERRORTYPE RNLInit(std::string &modelPath, float ratio, unsigned int bitDepth, RangeType rangeType,
                     unsigned int threadCount, ASMType asmType, unsigned int passes, unsigned int twoPassMode)
{
    std::cout << "RAISR [version]:\tRAISR Native Lib v" << RAISR_VERSION_MAJOR << "." << RAISR_VERSION_MINOR
              << " (" << RAISR_BACKEND << ")" << std::endl;
    std::cout << "-------------------------------------------\n";
    (void)threadCount;   // accepted and ignored: the GPU grid replaces the CPU row bands

    G.passes = 1; G.twoPassMode = 1;
    if (passes == 2) {
        G.passes = passes;
        G.twoPassMode = twoPassMode;
        std::cout << "--------------- running 2 pass ---------------\n";
    } else if (passes == 1 && twoPassMode == 2) {
        std::cout << "[RAISR WARNING] 1 pass with upscale in 2d pass, mode = 2 ignored !" << std::endl;
    } else if (passes != 1) {
        std::cout << "[RAISR ERROR] Only support passes 1 or 2. " << std::endl;
        return RNLErrorUndefined;
    }
    if (G.passes == 2 && G.twoPassMode != 1 && G.twoPassMode != 2) {
        std::cout << "[RAISR ERROR] Only support mode 1 or 2. " << std::endl;
        return RNLErrorBadParameter;
    }

    std::string hashtablePath = modelPath + "/" + "/filterbin_2";
    std::string strPath = modelPath + "/" + "/Qfactor_strbin_2";
    std::string cohPath = modelPath + "/" + "/Qfactor_cohbin_2";
    const std::string configPath = modelPath + "/" + "/config";

    const bool video = rangeType == VideoRange;
    if (bitDepth == 8) {
        hashtablePath += "_8"; strPath += "_8"; cohPath += "_8";
        G.lo = video ? 16 : 0; G.hi = video ? 235 : 255;
    } else if (bitDepth == 10) {
        hashtablePath += "_10"; strPath += "_10"; cohPath += "_10";
        G.lo = video ? 64 : 0; G.hi = video ? 940 : 1023;
    } else if (bitDepth == 16) {
        hashtablePath += "_16"; strPath += "_16"; cohPath += "_16";
        G.lo = 0; G.hi = 65535;
    } else {
        std::cout << "[RAISR ERROR] bit depth: " << bitDepth << "bits is NOT supported." << std::endl;
        return RNLErrorBadParameter;
    }
    if (!(ratio > 1.0f && ratio <= 2.0f)) {
        std::cout << "[RAISR ERROR] ratio: " << ratio << " is NOT supported." << std::endl;
        return RNLErrorBadParameter;
    }
    G.ratio = ratio;
    G.bitDepth = bitDepth;

    // asm mapping: which x86 path's output the GPU reproduces (DESIGN.md "asm mapping")
    switch ((int)asmType) {
    case AVX2:
        G.hashVariant = RAISR_HIP_HASH_AVX2;
        std::cout << "ASM Type: HIP gfx950 (AVX2-exact numerics)\n";
        break;
    case OpenCL:
    case OpenCLExternal:
        std::cout << "ASM Type: OpenCL requested, but OpenCL is not enabled.\n";
        return RNLErrorBadParameter;
    case AVX512_FP16:
        if (bitDepth == 8) {
            G.hashVariant = RAISR_HIP_HASH_FP16;
            std::cout << "ASM Type: HIP gfx950 (AVX512FP16-exact numerics)\n";
        } else {    // the binary16 pipeline overflows above 8-bit content (1023^2 > 65504)
            std::cout << "ASM Type: AVX512FP16 numerics requested, but they are defined for 8-bit content only.  Changing to AVX512\n";
            G.hashVariant = RAISR_HIP_HASH_AVX512;
            std::cout << "ASM Type: HIP gfx950 (AVX512-exact numerics)\n";
        }
        break;
    default:   // AVX512, HIP and out-of-range values (the reference also falls back to its best path)
        G.hashVariant = RAISR_HIP_HASH_AVX512;
        std::cout << "ASM Type: HIP gfx950 (AVX512-exact numerics)\n";
        break;
    }

    // config: "angle strength coherence patch"
    std::ifstream configFile(configPath);
    if (!configFile.is_open()) {
        std::cout << "[RAISR ERROR] Unable to open config file: " << configPath << std::endl;
        return RNLErrorBadParameter;
    }
    std::string line;
    std::getline(configFile, line);
    std::istringstream iss(line);
    std::vector<std::string> tokens{std::istream_iterator<std::string>{iss}, std::istream_iterator<std::string>{}};
    if (tokens.size() != 4) {
        std::cout << "[RAISR ERROR] configFile corrupted: " << configPath << std::endl;
        return RNLErrorBadParameter;
    }
    if (RNLErrorNone != parseUnsigned(&G.qAngle, tokens[0], configPath)) return RNLErrorBadParameter;
    if (RNLErrorNone != parseUnsigned(&G.qStrength, tokens[1], configPath)) return RNLErrorBadParameter;
    if (RNLErrorNone != parseUnsigned(&G.qCoherence, tokens[2], configPath)) return RNLErrorBadParameter;
    if (RNLErrorNone != parseUnsigned(&G.patchSize, tokens[3], configPath)) return RNLErrorBadParameter;
    if (G.patchSize != 11) {
        std::cout << "[RAISR ERROR] configFile corrupted: " << configPath << std::endl;
        return RNLErrorBadParameter;
    }
    // the GPU hash kernel is specialised for the 24x3x3 quantisation every shipped model uses
    if (G.qStrength != 3 || G.qCoherence != 3 || G.qAngle == 0 || G.qAngle * 9 > 255) {
        std::cout << "[RAISR ERROR] configFile corrupted: " << configPath << std::endl;
        return RNLErrorBadParameter;
    }

    if (RNLErrorNone != readTrainedData(hashtablePath, strPath, cohPath, 1, G.model[0])) return RNLErrorBadParameter;
    if (G.passes == 2 && RNLErrorNone != readTrainedData(hashtablePath, strPath, cohPath, 2, G.model[1]))
        return RNLErrorBadParameter;

    dropContext();
    int rc = raisr_hip_create(&G.ctx, G.device);
    if (rc != RAISR_HIP_OK) {
        std::cout << "[RAISR ERROR] HIP backend unavailable: " << raisr_hip_last_error() << std::endl;
        return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : RNLErrorUndefined;
    }
    for (unsigned p = 0; p < G.passes; p++) {
        const PassModel &M = G.model[p];
        rc = raisr_hip_set_model(G.ctx, (int)p, M.bank.data(), (int)M.hashkeys, (int)M.pixelTypes, M.qstr.data(), M.qcoh.data(), (int)G.qAngle);
        if (rc != RAISR_HIP_OK) {
            std::cout << "[RAISR ERROR] uploading model failed: " << raisr_hip_last_error() << std::endl;
            dropContext();
            return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : RNLErrorBadParameter;
        }
    }
    G.inited = true;
    return RNLErrorNone;
}

RNLERRORTYPE RNLSetRes(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                       VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb)
{
    (void)inCr; (void)inCb; (void)outCr; (void)outCb;
    if (!G.inited || !G.ctx || !inY || !outY) return RNLErrorBadParameter;
    raisr_hip_config cfg{};
    cfg.in_width = (int)inY->width; cfg.in_height = (int)inY->height;
    cfg.out_width = (int)outY->width; cfg.out_height = (int)outY->height;
    cfg.bits = (int)G.bitDepth;
    cfg.clamp_lo = G.lo; cfg.clamp_hi = G.hi;
    cfg.passes = (int)G.passes; cfg.two_pass_mode = (int)G.twoPassMode;
    cfg.hash_variant = G.hashVariant;
    cfg.blending = RAISR_HIP_BLEND_COUNT;
    cfg.use_pixel_type = G.ratio == 2.0f ? 1 : 0;     // gUsePixelType, Raisr.cpp:1477-1480
    cfg.tie_rule = RAISR_HIP_TIE_HALF_UP;
    const int rc = raisr_hip_configure(G.ctx, &cfg);
    if (rc != RAISR_HIP_OK) {
        std::cout << "[RAISR ERROR] set resolution failed: " << raisr_hip_last_error() << std::endl;
        return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : RNLErrorBadParameter;
    }
    G.resSet = true;
    return RNLErrorNone;
}

RNLERRORTYPE RNLProcess(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                        VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb, BlendingMode blendingMode)
{
    if (!inCr || !inCr->pData || !outCr || !outCr->pData || !inY || !inY->pData || !outY || !outY->pData)
        return RNLErrorBadParameter;
    if (!inCb || !inCb->pData || !outCb || !outCb->pData) return RNLErrorBadParameter;
    if (!G.inited || !G.resSet || !G.ctx) return RNLErrorBadParameter;
    if (blendingMode != CountOfBitsChanged && blendingMode != Randomness) return RNLErrorBadParameter;
    if (raisr_hip_set_blending(G.ctx, (int)blendingMode) != RAISR_HIP_OK) return RNLErrorBadParameter;
    const int rc = raisr_hip_process_host(G.ctx, inY->pData, inY->step, outY->pData, outY->step,
                                          inCr->pData, inCr->step, outCr->pData, outCr->step,
                                          inCb->pData, inCb->step, outCb->pData, outCb->step,
                                          (int)inCr->width, (int)inCr->height, (int)outCr->width, (int)outCr->height);
    if (rc != RAISR_HIP_OK) {
        std::cout << "[RAISR ERROR] process failed: " << raisr_hip_last_error() << std::endl;
        return RNLErrorUndefined;
    }
    return RNLErrorNone;
}

RNLERRORTYPE RNLSetOpenCLContext(void *context, void *deviceID, int platformIndex, int deviceIndex)
{
    (void)context; (void)deviceID; (void)platformIndex;
    G.device = deviceIndex < 0 ? 0 : deviceIndex;     // HIP device ordinal (vf_raisr `device=` option)
    return RNLErrorNone;
}

RNLERRORTYPE RNLDeinit()
{
    dropContext();
    G.inited = false;
    for (auto &m : G.model) { m.bank.clear(); m.bank.shrink_to_fit(); }
    return RNLErrorNone;
}

// ---- C ABI (RaisrHandler.cpp:11-60 in the reference: one-line forwards) -------------------------

extern "C" {

RNLERRORTYPE RNLHandler_Init(const char *modelPath, float ratio, unsigned int bitDepth, RangeType rangeType,
                             unsigned int threadCount, ASMType asmType, unsigned int passes, unsigned int twoPassMode)
{
    if (!modelPath) return RNLErrorBadParameter;
    std::string model = modelPath;
    return RNLInit(model, ratio, bitDepth, rangeType, threadCount, asmType, passes, twoPassMode);
}

RNLERRORTYPE RNLHandler_SetRes(VideoDataType *inY, VideoDataType *inU, VideoDataType *inV,
                               VideoDataType *outY, VideoDataType *outU, VideoDataType *outV)
{
    return RNLSetRes(inY, inU, inV, outY, outU, outV);
}

RNLERRORTYPE RNLHandler_Process(VideoDataType *inY, VideoDataType *inU, VideoDataType *inV,
                                VideoDataType *outY, VideoDataType *outU, VideoDataType *outV, BlendingMode blendingMode)
{
    return RNLProcess(inY, inU, inV, outY, outU, outV, blendingMode);
}

RNLERRORTYPE RNLHandler_SetOpenCLContext(void *context, void *device_id, int platformIndex, int deviceIndex)
{
    return RNLSetOpenCLContext(context, device_id, platformIndex, deviceIndex);
}

RNLERRORTYPE RNLHandler_Deinit(void)
{
    return RNLDeinit();
}

}  // extern "C"
