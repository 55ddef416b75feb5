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
#include <cstdio>
#include <cstdlib>
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
    raisr_hip_ctx *ctx = nullptr;                 // band 0 (the whole frame when the frame is not cut)
    // Horizontal bands (include/raisr_hip.h "Horizontal bands"): band k runs on its own context/stream so that
    // its uploads and downloads overlap the other bands' kernels inside one synchronous RNLProcess call.
    std::vector<raisr_hip_ctx *> extra;           // contexts of bands 1..K-1
    std::vector<raisr_hip_band> yBands, cBands;   // luma / chroma plans (same count; cBands empty = chroma on band 0)
    int chromaInH = 0, chromaOutH = 0;
    // plane geometry RNLSetRes configured: RNLProcess refuses frames that differ (the copies are asynchronous DMAs)
    unsigned geo[6][2] = {};                      // {width, height} of inY, inCr, inCb, outY, outCr, outCb
    raisr_hip_config cfg{};                       // what RNLSetRes configured (the async ring is configured with the same)
    // asynchronous frames (RNLHandler_Submit / _Collect): a ring of `asyncDepth` lanes over the same models and geometry
    raisr_hip_stream *ring = nullptr;
    unsigned asyncDepth = 0;                      // RNLHandler_SetAsyncDepth; 0 = no ring
    bool deviceChosen = false;                    // RNLSetOpenCLContext named a device; otherwise RAISR_HIP_DEVICE / 0
    std::vector<int> devices;                     // RNLSetDeviceList: GPUs of the asynchronous ring
    bool devicesSet = false;                      // RNLSetDeviceList was called: its list ("" = the handler's one device) wins over RAISR_HIP_DEVICES
    bool external = false;                        // asm = HIPExternal: plane pointers are device pointers
    void *externalStream = nullptr;               // caller's hipStream_t for external frames (NULL: own stream + wait)
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

// ---- page-locked caller planes ------------------------------------------------------------------
// A copy on page-locked memory is a DMA the copy engines run next to the kernels.  The supported way to get there is memory the
// host takes from RNLHandler_HostAlloc (FFmpeg: the buffer pools of ffmpeg/vf_raisr_hip.diff); the device layer recognises it
// per plane (csrc/host_copy.h) and carries pageable planes through page-locked bounce memory of its own -- nothing here touches
// memory the library does not own.
//
// RAISR_HIP_PIN=1 (OPT-IN) additionally page-locks ordinary planes on first sight and remembers the registration (bounded,
// least-recently-used region dropped; everything is unlocked in RNLDeinit).  The host then promises that every plane buffer it
// hands over stays allocated until RNLDeinit: a registration outlives a free() of the memory under it, and whatever the
// allocator puts at that address next is treated as page-locked by the runtime.  That is why this is not the default.  The
// runtime refuses a copy whose host range is only PARTLY inside a registered range -- so the ranges (exact plane extents, not
// rounded to pages) are kept disjoint, and a plane that overlaps registered ranges without lying inside one (a band of rows
// registered first, the whole plane later) gets the union registered as ONE range, after waiting for the frames in flight.
void quiesceDevice();

struct PinCache {
    struct Ent { uintptr_t lo, hi; uint64_t stamp; bool ours; };     // [lo, hi): one registered region (ours: we unlock it)
    std::vector<Ent> ents;                                            // pairwise disjoint
    std::vector<std::pair<uintptr_t, uintptr_t>> refused;             // ranges the runtime would not register: not asked again
    uint64_t clock = 0;
    size_t cap = 48;                                  // > 2 x 3 planes x the deepest pool vf_raisr sees in practice
    int enabled = -1;

    bool on()
    {
        if (enabled < 0) { const char *e = std::getenv("RAISR_HIP_PIN"); enabled = (e && std::atoi(e) == 1) ? 1 : 0; }
        return enabled == 1;
    }
    int debug = -1;
    void log(const char *what, uintptr_t lo, uintptr_t hi, int rc)
    {
        if (debug < 0) debug = std::getenv("RAISR_HIP_PIN_DEBUG") ? 1 : 0;
        if (debug) std::fprintf(stderr, "[raisr pin] %s [%#zx, %#zx) %zu KB -> %d (%zu regions)\n", what, (size_t)lo, (size_t)hi, (size_t)(hi - lo) >> 10, rc, ents.size());
    }
    void drop(size_t k)
    {
        if (ents[k].ours) { const int rc = raisr_hip_host_unregister((void *)ents[k].lo); log("unregister", ents[k].lo, ents[k].hi, rc); }
        ents.erase(ents.begin() + (long)k);
    }
    void pin(const void *p, size_t bytes)
    {
        if (!on() || !p || !bytes) return;
        // the EXACT byte range: the runtime keys a registration by the range it was given (it locks the pages underneath), so a
        // neighbouring buffer that merely shares the first or last page is not "partly registered"
        uintptr_t lo = (uintptr_t)p, hi = (uintptr_t)p + bytes;
        std::vector<size_t> hit;
        for (size_t i = 0; i < ents.size(); i++)
            if (ents[i].hi > lo && ents[i].lo < hi) hit.push_back(i);
        if (hit.size() == 1 && ents[hit[0]].lo <= lo && hi <= ents[hit[0]].hi) { ents[hit[0]].stamp = ++clock; return; }
        if (hit.empty() && bytes < (size_t)64 * 1024) return;             // a small plane on pages nobody locked: the pageable path is fine
        for (const auto &r : refused) if (r.first == lo && r.second == hi) return;
        if (!hit.empty()) {
            // the plane straddles registered pages: replace the regions it touches by their union with the plane
            for (size_t i : hit) if (!ents[i].ours) return;               // somebody else's registration: nothing we can merge
            quiesceDevice();
            for (size_t k = hit.size(); k-- > 0;) {
                lo = ents[hit[k]].lo < lo ? ents[hit[k]].lo : lo;
                hi = ents[hit[k]].hi > hi ? ents[hit[k]].hi : hi;
                drop(hit[k]);
            }
        }
        while (ents.size() >= cap) {                                      // drop the least recently used region
            size_t k = 0;
            for (size_t i = 1; i < ents.size(); i++) if (ents[i].stamp < ents[k].stamp) k = i;
            quiesceDevice();
            drop(k);
        }
        const int rc = raisr_hip_host_register((void *)lo, hi - lo);
        log(hit.empty() ? "register" : "register union", lo, hi, rc);
        if (rc == RAISR_HIP_OK) ents.push_back({lo, hi, ++clock, true});
        else if (rc == RAISR_HIP_ESTATE) ents.push_back({lo, hi, ++clock, false});      // page-locked by the host itself
        else if (refused.size() < 256) refused.emplace_back(lo, hi);
    }
    void clear()
    {
        for (Ent &e : ents) if (e.ours) (void)raisr_hip_host_unregister((void *)e.lo);
        ents.clear(); refused.clear();
        enabled = -1;                                   // the next session reads RAISR_HIP_PIN again
    }
} gPins;

void pinPlanes(VideoDataType *const pl[6])
{
    if (!gPins.on()) return;
    for (int i = 0; i < 6; i++)
        if (pl[i] && pl[i]->pData) gPins.pin(pl[i]->pData, (size_t)pl[i]->step * pl[i]->height);
}

// everything this library has enqueued is done (frames of the ring stay "in flight" for the caller: Collect returns at once)
void quiesceDevice()
{
    if (G.ctx) (void)raisr_hip_synchronize(G.ctx);
    for (raisr_hip_ctx *c : G.extra) (void)raisr_hip_synchronize(c);
    if (G.ring) (void)raisr_hip_stream_quiesce(G.ring);
}

void dropRing()
{
    if (G.ring) { raisr_hip_stream_destroy(G.ring); G.ring = nullptr; }      // waits for the frames in flight
}

void dropExtraBands()
{
    for (raisr_hip_ctx *c : G.extra) raisr_hip_destroy(c);
    G.extra.clear(); G.yBands.clear(); G.cBands.clear();
}

void dropContext()
{
    dropRing();
    dropExtraBands();
    if (G.ctx) { raisr_hip_destroy(G.ctx); G.ctx = nullptr; }
    G.resSet = false;
}

int uploadModels(raisr_hip_ctx *ctx)
{
    for (unsigned p = 0; p < G.passes; p++) {
        const PassModel &M = G.model[p];
        const int rc = raisr_hip_set_model(ctx, (int)p, M.bank.data(), (int)M.hashkeys, (int)M.pixelTypes, M.qstr.data(), M.qcoh.data(), (int)G.qAngle);
        if (rc != RAISR_HIP_OK) return rc;
    }
    return RAISR_HIP_OK;
}

// How many bands RNLProcess cuts a frame into: RAISR_HIP_BANDS=n (default 1 = never cut).  Measured on
// 1080p->4K yuv420p: the host path is bound by the plane copies themselves (0.54 ms/frame uncut; 0.52 ms with two
// bands, slower with more: every band adds launches and padding rows), so cutting is left to callers that want it --
// page-locked planes, or one band per GPU through the device layer.
int wantedBands(int outHeight)
{
    if (const char *e = std::getenv("RAISR_HIP_BANDS")) {
        const int n = std::atoi(e);
        if (n >= 1) return n > 16 ? 16 : n;
    }
    (void)outHeight;
    return 1;
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

    // A re-initialisation starts from scratch: whatever an earlier RNLInit left behind (device context, geometry,
    // models) must not survive a failure below and run with the new, mismatched parameters.
    dropContext();
    G.inited = false;
    if (!G.deviceChosen) {                       // unmodified vf_raisr only names a device for asm=opencl
        if (const char *e = std::getenv("RAISR_HIP_DEVICE")) { const int d = std::atoi(e); G.device = d < 0 ? 0 : d; }
    }

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
    G.external = (int)asmType == HIPExternal;
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
        if (bitDepth <= 10) {   // 10-bit: the reference runs its binary16 path too (Convert_8u16f_10bit, NF_10), overflows and all
            G.hashVariant = RAISR_HIP_HASH_FP16;
            std::cout << "ASM Type: HIP gfx950 (AVX512FP16-exact numerics)\n";
        } else {    // 16-bit samples are not exact in binary16
            std::cout << "ASM Type: AVX512FP16 numerics requested, but 16-bit samples are not exact in binary16.  Changing to AVX512\n";
            G.hashVariant = RAISR_HIP_HASH_AVX512;
            std::cout << "ASM Type: HIP gfx950 (AVX512-exact numerics)\n";
        }
        break;
    case HIPExternal:
        G.hashVariant = RAISR_HIP_HASH_AVX512;
        std::cout << "ASM Type: HIP gfx950 (AVX512-exact numerics), device-resident frames\n";
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
    rc = uploadModels(G.ctx);
    if (rc != RAISR_HIP_OK) {
        std::cout << "[RAISR ERROR] uploading model failed: " << raisr_hip_last_error() << std::endl;
        dropContext();
        return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : RNLErrorBadParameter;
    }
    G.inited = true;
    return RNLErrorNone;
}

RNLERRORTYPE RNLSetRes(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                       VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb)
{
    if (!G.inited || !G.ctx || !inY || !outY) return RNLErrorBadParameter;
    {
        VideoDataType *pl[6] = {inY, inCr, inCb, outY, outCr, outCb};
        for (int i = 0; i < 6; i++) { G.geo[i][0] = pl[i] ? pl[i]->width : 0; G.geo[i][1] = pl[i] ? pl[i]->height : 0; }
    }
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

    // frames submitted and not yet collected would be lost with the ring: the caller collects first (the reference has no frames
    // in flight; this keeps Submit / Collect pairs intact)
    if (G.ring && raisr_hip_stream_in_flight(G.ring) > 0) {
        std::cout << "[RAISR ERROR] RNLSetRes with " << raisr_hip_stream_in_flight(G.ring) << " frame(s) in flight: collect them first" << std::endl;
        return RNLErrorBadParameter;
    }
    G.cfg = cfg;
    // band plan: luma with the pass count's padding, chroma (cheap upscale only) with its own
    dropRing();
    dropExtraBands();
    G.resSet = false;
    const int want = G.external ? 1 : wantedBands(cfg.out_height);
    std::vector<raisr_hip_band> yb((size_t)want), cb((size_t)want);
    int K = want > 1 ? raisr_hip_plan_bands(cfg.in_height, cfg.out_height, cfg.passes, want, yb.data()) : 1;
    if (K < 1) K = 1;
    G.chromaInH = inCr ? (int)inCr->height : 0; G.chromaOutH = outCr ? (int)outCr->height : 0;
    bool chromaBanded = false;
    if (K > 1 && G.chromaInH > 0 && G.chromaOutH > 0)
        chromaBanded = raisr_hip_plan_bands(G.chromaInH, G.chromaOutH, 0, K, cb.data()) == K;

    auto failed = [&](int rc) {
        std::cout << "[RAISR ERROR] set resolution failed: " << raisr_hip_last_error() << std::endl;
        dropExtraBands();
        return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : RNLErrorBadParameter;
    };
    if (K > 1) {
        for (int k = 0; k < K; k++) {
            raisr_hip_ctx *ctx = G.ctx;
            if (k > 0) {
                ctx = nullptr;
                int rc = raisr_hip_create(&ctx, G.device);
                if (rc == RAISR_HIP_OK) { G.extra.push_back(ctx); rc = uploadModels(ctx); }
                if (rc != RAISR_HIP_OK) return failed(rc);
            }
            if (k > 0 && !(std::getenv("RAISR_HIP_BAND_CHAIN") && std::atoi(std::getenv("RAISR_HIP_BAND_CHAIN")) == 0)) {
                // band k's kernels after band k-1's: the bands' kernels then run back to back and every band's download
                // overlaps the next band's kernels (unordered, the bands share the GPU and all finish -- and download -- together)
                raisr_hip_ctx *prev = k == 1 ? G.ctx : G.extra[(size_t)k - 2];
                const int rc = raisr_hip_set_after(ctx, prev);
                if (rc != RAISR_HIP_OK) return failed(rc);
            }
            raisr_hip_config sub = cfg;
            sub.in_height = yb[(size_t)k].in_row_count; sub.out_height = yb[(size_t)k].out_row_count;
            const int rc = raisr_hip_configure(ctx, &sub);
            if (rc != RAISR_HIP_OK) return failed(rc);
        }
        G.yBands.assign(yb.begin(), yb.begin() + K);
        if (chromaBanded) G.cBands.assign(cb.begin(), cb.begin() + K);
    } else {
        const int rc = raisr_hip_configure(G.ctx, &cfg);
        if (rc != RAISR_HIP_OK) return failed(rc);
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
    {   // the frame must have the geometry RNLSetRes sized the device planes for, and rows at least one line long
        VideoDataType *pl[6] = {inY, inCr, inCb, outY, outCr, outCb};
        const unsigned bps = G.bitDepth == 8 ? 1u : 2u;
        for (int i = 0; i < 6; i++) {
            if (i == 2 || i == 5) {            // RNLSetRes never looked at the Cb planes: they must match Cr
                if (pl[i]->width != pl[i - 1]->width || pl[i]->height != pl[i - 1]->height) return RNLErrorBadParameter;
            } else if (pl[i]->width != G.geo[i][0] || pl[i]->height != G.geo[i][1]) return RNLErrorBadParameter;
            const unsigned per = (i != 0 && i != 3 && (pl[i]->bitShift & RAISR_HIP_INTERLEAVED2)) ? 2u : 1u;
            if ((uint64_t)pl[i]->step < (uint64_t)pl[i]->width * bps * per) return RNLErrorBadParameter;
        }
    }
    auto failed = [&]() {
        std::cout << "[RAISR ERROR] process failed: " << raisr_hip_last_error() << std::endl;
        return RNLErrorUndefined;
    };
    const size_t K = G.yBands.size();
    if (!G.external) {
        if ((inCr->bitShift | inCb->bitShift | outCr->bitShift | outCb->bitShift) & RAISR_HIP_INTERLEAVED2) return RNLErrorBadParameter;   // device frames only
        VideoDataType *pl[6] = {inY, inCr, inCb, outY, outCr, outCb};
        pinPlanes(pl);
        // whole frames on one context: the rows of the last pass go back in three ranges while the next range is computed
        // (1080p -> 4K, page-locked planes: 2.3 k -> 2.55 k frames/s through this synchronous entry; pageable planes: the unpacking
        // of a range from the bounce memory overlaps the next range's kernels); RAISR_HIP_CHUNKS overrides
        if (K == 0) (void)raisr_hip_set_chunks(G.ctx, 3);
    }
    if (G.external) {
        // planes are device pointers: RAISR on Y and the cheap upscale of both chroma planes without leaving HBM
        if (inCr->step != inCb->step || outCr->step != outCb->step) return RNLErrorBadParameter;
        if (raisr_hip_set_blending(G.ctx, (int)blendingMode) != RAISR_HIP_OK) return RNLErrorBadParameter;
        // NV12 / P010 surfaces: both chroma descriptors flag one interleaved plane (RaisrDefaults.h) -- all four or none
        const unsigned il = (inCr->bitShift & inCb->bitShift & outCr->bitShift & outCb->bitShift) & RAISR_HIP_INTERLEAVED2;
        if (((inCr->bitShift | inCb->bitShift | outCr->bitShift | outCb->bitShift) & RAISR_HIP_INTERLEAVED2) && !il) return RNLErrorBadParameter;
        // bitShift (low bits): samples of the surface are MSB-aligned by that many bits (P010: 6).  Read from the INPUT descriptors
        // only, as the reference does (Raisr.cpp:1313-1348: inY / inCr / inCb->bitShift); a caller that leaves the output
        // descriptors' field at 0 gets the input's alignment on the output, as there.
        const unsigned sh = inY->bitShift & 0xFFu;
        for (const VideoDataType *p : {inCr, inCb}) if ((p->bitShift & 0xFFu) != sh) return RNLErrorBadParameter;
        if (raisr_hip_set_sample_shift(G.ctx, (int)sh) != RAISR_HIP_OK) return RNLErrorBadParameter;
        int rc = raisr_hip_process_frame_device_ex(G.ctx, inY->pData, inY->step, outY->pData, outY->step,
                                                   inCr->pData, inCb->pData, inCr->step, outCr->pData, outCb->pData, outCr->step,
                                                   (int)inCr->width, (int)inCr->height, (int)outCr->width, (int)outCr->height, il ? 2 : 1, G.externalStream);
        if (rc == RAISR_HIP_OK && !G.externalStream) rc = raisr_hip_synchronize(G.ctx);
        return rc != RAISR_HIP_OK ? failed() : RNLErrorNone;
    }
    if (K <= 1) {
        if (raisr_hip_set_blending(G.ctx, (int)blendingMode) != RAISR_HIP_OK) return RNLErrorBadParameter;
        const int rc = raisr_hip_process_host(G.ctx, inY->pData, inY->step, outY->pData, outY->step,
                                              inCr->pData, inCr->step, outCr->pData, outCr->step,
                                              inCb->pData, inCb->step, outCb->pData, outCb->step,
                                              (int)inCr->width, (int)inCr->height, (int)outCr->width, (int)outCr->height);
        return rc != RAISR_HIP_OK ? failed() : RNLErrorNone;
    }
    // the planes must still have the geometry the bands were planned for
    if ((int)inCr->height != G.chromaInH || (int)outCr->height != G.chromaOutH) return RNLErrorBadParameter;
    // stage 1 (upload + kernels) for every band first, then the downloads: a download into pageable memory blocks
    // this thread until its band's kernels are done, and meanwhile the later bands' kernels keep the GPU busy
    int rc = RAISR_HIP_OK;
    for (int stage = 1; stage <= 2 && rc == RAISR_HIP_OK; stage++) {
        for (size_t k = 0; k < K && rc == RAISR_HIP_OK; k++) {
            raisr_hip_ctx *ctx = k == 0 ? G.ctx : G.extra[k - 1];
            if (stage == 1 && raisr_hip_set_blending(ctx, (int)blendingMode) != RAISR_HIP_OK) return RNLErrorBadParameter;
            const raisr_hip_band &y = G.yBands[k];
            raisr_hip_rows rows{y.keep_begin - y.out_row_begin, y.keep_count, 0, 0, stage};
            const unsigned char *iu = nullptr, *iv = nullptr;
            unsigned char *ou = nullptr, *ov = nullptr;
            int cih = 0, coh = 0;
            if (!G.cBands.empty()) {
                const raisr_hip_band &c = G.cBands[k];
                rows.c_skip = c.keep_begin - c.out_row_begin; rows.c_keep = c.keep_count;
                iu = inCr->pData + (size_t)c.in_row_begin * inCr->step; iv = inCb->pData + (size_t)c.in_row_begin * inCb->step;
                ou = outCr->pData + (size_t)c.keep_begin * outCr->step; ov = outCb->pData + (size_t)c.keep_begin * outCb->step;
                cih = c.in_row_count; coh = c.out_row_count;
            } else if (k == 0) {            // chroma too small to cut: band 0 does the whole planes
                rows.c_skip = 0; rows.c_keep = (int)outCr->height;
                iu = inCr->pData; iv = inCb->pData; ou = outCr->pData; ov = outCb->pData;
                cih = (int)inCr->height; coh = (int)outCr->height;
            }
            rc = raisr_hip_process_host_async(ctx, inY->pData + (size_t)y.in_row_begin * inY->step, inY->step,
                                              outY->pData + (size_t)y.keep_begin * outY->step, outY->step,
                                              iu, inCr->step, ou, outCr->step, iv, inCb->step, ov, outCb->step,
                                              (int)inCr->width, cih, (int)outCr->width, coh, &rows);
        }
    }
    // always drain every band, also after an error, so that no transfer into the caller's planes is left in flight
    for (size_t k = 0; k < K; k++) {
        const int r2 = raisr_hip_synchronize(k == 0 ? G.ctx : G.extra[k - 1]);
        if (rc == RAISR_HIP_OK) rc = r2;
    }
    return rc != RAISR_HIP_OK ? failed() : RNLErrorNone;
}

RNLERRORTYPE RNLSetOpenCLContext(void *context, void *deviceID, int platformIndex, int deviceIndex)
{
    (void)deviceID; (void)platformIndex;
    G.externalStream = context;                       // asm = HIPExternal: the caller's hipStream_t (or NULL)
    G.device = deviceIndex < 0 ? 0 : deviceIndex;     // HIP device ordinal (vf_raisr `device=` option)
    G.deviceChosen = true;
    return RNLErrorNone;
}

RNLERRORTYPE RNLDeinit()
{
    dropContext();
    G.inited = false;
    for (auto &m : G.model) { m.bank.clear(); m.bank.shrink_to_fit(); }
    // what RNLSetOpenCLContext stored belongs to the session that ends here: the caller may destroy its stream now, and the
    // next RNLInit without a SetOpenCLContext call goes back to RAISR_HIP_DEVICE / device 0
    G.externalStream = nullptr;
    G.deviceChosen = false;
    G.device = 0;
    G.asyncDepth = 0;
    G.devices.clear();
    G.devicesSet = false;
    gPins.clear();
    return RNLErrorNone;
}

// ---- page-locked frame memory (extension) -----------------------------------------------------------------------------------
void *RNLHostAlloc(size_t bytes)
{
    if (!bytes) return nullptr;
    return raisr_hip_host_alloc(bytes);
}

void RNLHostFree(void *p) { raisr_hip_host_free(p); }

// ---- asynchronous frames (extension; the reference's Process is synchronous, Raisr.cpp:1294-1397) ----------------------------
// RNLSubmit enqueues a frame on the next lane of a ring (upload, kernels, download: all asynchronous on page-locked planes) and
// returns; RNLCollect waits for the OLDEST submitted frame.  Same validation and the same bits as RNLProcess; the caller keeps
// every plane valid and untouched between a frame's Submit and its Collect.
static bool ringDevices(std::vector<int> &devs);

RNLERRORTYPE RNLSetAsyncDepth(unsigned int depth)
{
    if (depth > RAISR_HIP_STREAM_MAX_DEPTH) return RNLErrorBadParameter;           // what the ring builds; a larger request is refused, not clamped
    if (G.ring && raisr_hip_stream_in_flight(G.ring) > 0) return RNLErrorBadParameter;      // collect first
    if (depth) {            // a malformed RAISR_HIP_DEVICES is an error HERE (with its message), not a capacity of 0 later
        std::vector<int> devs;
        if (!ringDevices(devs)) return RNLErrorBadParameter;
    }
    dropRing();
    G.asyncDepth = depth;
    return RNLErrorNone;
}

RNLERRORTYPE RNLSetDeviceList(const char *devices)
{
    if (!devices) return RNLErrorBadParameter;
    if (G.ring && raisr_hip_stream_in_flight(G.ring) > 0) return RNLErrorBadParameter;      // collect first
    int list[RAISR_HIP_STREAM_MAX_DEVICES];
    const int n = raisr_hip_parse_device_list(devices, list, RAISR_HIP_STREAM_MAX_DEVICES);
    if (n < 0) return RNLErrorBadParameter;
    dropRing();
    G.devices.assign(list, list + n);
    G.devicesSet = true;
    return RNLErrorNone;
}

// the ring's GPUs: RNLSetDeviceList (an empty list there = the handler's one device, whatever the environment says), else
// RAISR_HIP_DEVICES, else the handler's one device; false: RAISR_HIP_DEVICES is malformed
static bool ringDevices(std::vector<int> &devs)
{
    devs = G.devices;
    if (devs.empty() && !G.devicesSet) {
        if (const char *e = std::getenv("RAISR_HIP_DEVICES")) {
            int list[RAISR_HIP_STREAM_MAX_DEVICES];
            const int n = raisr_hip_parse_device_list(e, list, RAISR_HIP_STREAM_MAX_DEVICES);
            if (n < 0) { std::cout << "[RAISR ERROR] RAISR_HIP_DEVICES=" << e << ": not a list of HIP devices of this machine" << std::endl; return false; }
            devs.assign(list, list + n);
        }
    }
    if (devs.empty()) devs.push_back(G.device);
    return true;
}

int RNLAsyncCapacity()
{
    if (G.ring) return raisr_hip_stream_depth(G.ring);
    std::vector<int> devs;
    if (G.asyncDepth == 0 || !ringDevices(devs)) return 0;
    return (int)(devs.size() * G.asyncDepth);
}

static RNLERRORTYPE checkFrame(VideoDataType *const pl[6])
{
    for (int i = 0; i < 6; i++) if (!pl[i] || !pl[i]->pData) return RNLErrorBadParameter;
    if (!G.inited || !G.resSet || !G.ctx) return RNLErrorBadParameter;
    const unsigned bps = G.bitDepth == 8 ? 1u : 2u;
    for (int i = 0; i < 6; i++) {
        if (i == 2 || i == 5) {
            if (pl[i]->width != pl[i - 1]->width || pl[i]->height != pl[i - 1]->height) return RNLErrorBadParameter;
        } else if (pl[i]->width != G.geo[i][0] || pl[i]->height != G.geo[i][1]) return RNLErrorBadParameter;
        if ((uint64_t)pl[i]->step < (uint64_t)pl[i]->width * bps) return RNLErrorBadParameter;
        if (pl[i]->bitShift & RAISR_HIP_INTERLEAVED2) return RNLErrorBadParameter;     // interleaved chroma: device frames only (RNLProcess, asm = HIPExternal)
    }
    return RNLErrorNone;
}

RNLERRORTYPE RNLSubmit(VideoDataType *inY, VideoDataType *inCr, VideoDataType *inCb,
                       VideoDataType *outY, VideoDataType *outCr, VideoDataType *outCb, BlendingMode blendingMode)
{
    VideoDataType *pl[6] = {inY, inCr, inCb, outY, outCr, outCb};
    const RNLERRORTYPE ok = checkFrame(pl);
    if (ok != RNLErrorNone) return ok;
    if (blendingMode != CountOfBitsChanged && blendingMode != Randomness) return RNLErrorBadParameter;
    if (G.external || G.asyncDepth == 0) return RNLErrorBadParameter;           // device-pointer frames are stream-ordered already
    auto failed = [&](const char *what) {
        std::cout << "[RAISR ERROR] " << what << ": " << raisr_hip_last_error() << std::endl;
        return RNLErrorUndefined;
    };
    if (!G.ring) {
        std::vector<int> devs;
        if (!ringDevices(devs)) return RNLErrorBadParameter;
        int rc = raisr_hip_stream_create_multi(&G.ring, devs.data(), (int)devs.size(), (int)G.asyncDepth);
        if (rc != RAISR_HIP_OK) { G.ring = nullptr; return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : failed("async ring"); }
        for (unsigned p = 0; p < G.passes && rc == RAISR_HIP_OK; p++) {
            const PassModel &M = G.model[p];
            rc = raisr_hip_stream_set_model(G.ring, (int)p, M.bank.data(), (int)M.hashkeys, (int)M.pixelTypes, M.qstr.data(), M.qcoh.data(), (int)G.qAngle);
        }
        if (rc == RAISR_HIP_OK) rc = raisr_hip_stream_configure(G.ring, &G.cfg);
        if (rc != RAISR_HIP_OK) { dropRing(); return rc == RAISR_HIP_ENOMEM ? RNLErrorInsufficientResources : failed("async ring set-up"); }
    }
    if (raisr_hip_stream_in_flight(G.ring) >= raisr_hip_stream_depth(G.ring)) return RNLErrorInsufficientResources;   // ring full: collect first
    pinPlanes(pl);
    if (raisr_hip_stream_set_blending(G.ring, (int)blendingMode) != RAISR_HIP_OK) return RNLErrorBadParameter;
    const int rc = raisr_hip_stream_submit(G.ring, inY->pData, inY->step, outY->pData, outY->step,
                                           inCr->pData, inCr->step, outCr->pData, outCr->step,
                                           inCb->pData, inCb->step, outCb->pData, outCb->step,
                                           (int)inCr->width, (int)inCr->height, (int)outCr->width, (int)outCr->height);
    return rc != RAISR_HIP_OK ? failed("submit failed") : RNLErrorNone;
}

RNLERRORTYPE RNLCollect()
{
    if (!G.ring || raisr_hip_stream_in_flight(G.ring) == 0) return RNLErrorBadParameter;
    if (raisr_hip_stream_collect(G.ring) != RAISR_HIP_OK) {
        std::cout << "[RAISR ERROR] collect failed: " << raisr_hip_last_error() << std::endl;
        return RNLErrorUndefined;
    }
    return RNLErrorNone;
}

int RNLFramesInFlight() { return G.ring ? raisr_hip_stream_in_flight(G.ring) : 0; }

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

RNLERRORTYPE RNLHandler_SetAsyncDepth(unsigned int depth) { return RNLSetAsyncDepth(depth); }
RNLERRORTYPE RNLHandler_SetDeviceList(const char *devices) { return RNLSetDeviceList(devices); }
int RNLHandler_AsyncCapacity(void) { return RNLAsyncCapacity(); }

RNLERRORTYPE RNLHandler_Submit(VideoDataType *inY, VideoDataType *inU, VideoDataType *inV,
                               VideoDataType *outY, VideoDataType *outU, VideoDataType *outV, BlendingMode blendingMode)
{
    return RNLSubmit(inY, inU, inV, outY, outU, outV, blendingMode);
}

RNLERRORTYPE RNLHandler_Collect(void) { return RNLCollect(); }

int RNLHandler_FramesInFlight(void) { return RNLFramesInFlight(); }

void *RNLHandler_HostAlloc(size_t bytes) { return RNLHostAlloc(bytes); }

void RNLHandler_HostFree(void *p) { RNLHostFree(p); }

}  // extern "C"
