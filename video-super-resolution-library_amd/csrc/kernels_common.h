// kernels_common.h -- constants, per-pass parameters, XCD-aware tile order, LDS barrier, tile staging
// Included by device_abi.hip inside its anonymous namespace, in the order given there (gfx950 only; built with
// -ffp-contract=off and without fast-math: every floating-point operation is ONE IEEE operation of the cited reference line).
#pragma once

constexpr int kTaps = 121;
constexpr int kTapsPad = 128;
constexpr int kMargin = 6;       // gLoopMargin, Raisr.cpp:1574
constexpr int kBlobHeader = 64;  // bytes

struct GaussW {
    float wT[11][12];            // wT[k][i] = weight of patch row i, column k (column-major for the k-outer loop)
};

struct PassParams {
    int W, H;                    // plane size of this pass
    int lr_pitch;                // LR plane pitch in u16 elements
    int hash_pitch;              // hash plane pitch in bytes (u8 elements)
    int hr_pitch;                // HR plane pitch in floats
    float lo, hi;                // accept-test / clamp limits as float
    int ilo, ihi;
    int a_begin, a_end;          // columns hashed with the AVX-512 flavour
    int b_begin, b_end;          // columns hashed with the AVX2 flavour (after the AVX-512 one)
    int c_final;                 // first column that is never filtered
    int ov_begin, ov_end;        // columns hashed twice (AVX-512 flavour, then AVX2): second hash lives in hash2
    const uint8_t* hash2;        // [H][16] second hash of the overlap columns (column c -> c - ov_begin)
    int pixel_types;             // 4 (ratio 2) or 1
    int randomness;              // 1: BlendingMode Randomness (tail re-hash candidate replaces, never keeps, the first)
    float qangle, qs0, qs1, qc0, qc1;
    const float* bank;           // [hash][type][128]
    int bank_bytes;              // size of the fp32 bank (buffer-descriptor range)
    const uint2* tab14;          // [128]: rcp14 {C0,C1}[64], rsqrt14 {C0,C1}[64]
    const uint16_t* lut_legacy;  // rcp[2048], rsqrt[2048]
    int write_hash;              // fused kernel: also write the hash plane (introspection for tests)
    const uint8_t* c1tab;        // class-1 (exactly one-dimensional windows) sign table of kernels_hash_certify.h, or null: class switched off
    int c1_ok;                   // bit 0 / 1: this model's coherence thresholds admit the class-1 rule in the AVX-512 / AVX2 flavour
    unsigned* cert_stats;        // certified-hash kernel: {pixels sent to the exact path, certified-but-wrong, zone pixels, tiles with a list, overflowed tiles, tiles, flat tiles, -} or null
    int cert_check;              // 1: every pixel also takes the exact path and certified buckets are compared with it (tests)
    int zero_bucket[2];          // bucket of the all-zero tensor in the AVX-512 / AVX2 flavour (flat windows), from the exact device code
    const float* gauss_dev;      // GaussW::wT as a device array [11][12] (per-lane weights of the 16-lane exact tensor)
    const float* bank_lm;        // the bank once more, lane-major: row r = bucket * pixel_types + type holds, at float (ch >> 2) * 64 + l * 4 + (ch & 3),
                                 // tap 16 ch + l -- the four (eight) coefficients of zmm lane l are one (two) 16-byte loads (k_lane_major_bank)
    const uint32_t* asym;        // symmetric filter stage (filter_phase<.., SYM>): bitmap [32 words] of the (bucket * pixel_types + type) bank rows
                                 // that are NOT palindromic (f[k] != f[120-k] for a k <= 56), or null when every row is
    // frame batches (raisr_hip_process_y_device_batch): blockIdx.z = frame; plane f of a batch starts f * zs_* ELEMENTS after plane 0
    // (all 0 for a single frame).  Only the production kernels (k_hashfilter_ac, k_blend, k_hashfilter16, k_blend16) read these.
    size_t zs_lr, zs_hr, zs_hash, zs_out;
    int out_shift;               // blend kernels: stored sample = value << out_shift (MSB-aligned device frames, see ResizeParams); 0 otherwise
    int tile_y0;                 // first tile row of this launch (k_hashfilter_ac / k_blend launched on a range of tile rows: the host
                                 // path pipelines the download of finished rows with the kernels of the next rows)
};

// Deferred exact path of k_hashfilter_ac<.., DEFER> (kernels_hash_certify.h: hash_phase_defer; kernels_fix.h: k_fix_ac).  The pixels
// whose bucket the certified hash stage cannot vouch for are filtered with their approximate bucket and listed here, one
// region per (tile, wave) -- no atomics, no workgroup barrier; k_fix_ac, stream-ordered between the main kernel and k_blend,
// computes their exact bucket and overwrites the HR value of those whose bucket was wrong.
constexpr unsigned kWaveCap = 64;   // entries per wave region (of the wave's 256 pixels); a wave with more takes the all-exact code for its rows itself
struct FixAc {
    uint8_t* counts;             // [frame][tile][4]: entries in wave w's region, 0xFF = the wave ran the all-exact code (nothing listed)
    uint16_t* entries;           // [frame][tile][4][kWaveCap]: row within the wave's four (2 bits) | column within the tile (6) | approximate bucket << 8
    int tiles_x;                 // tiles per tile row of the plane (tile id = tile row * tiles_x + tile column)
    unsigned zs_tiles;           // frame batches: tiles per frame
};

#ifdef RAISR_HIP_DEV
// development builds: wave-cycles per phase of the fused kernel (s_memtime at the phase boundaries of every wave, summed over the launch;
// the instrumentation itself costs ~10 % -- read the shares, not the totals).  raisr_hip_dev_phase_stats() reads and clears them.
__device__ unsigned long long g_phase_cycles[8];
__device__ __forceinline__ void phase_mark(unsigned long long& t, int k, unsigned tid)
{
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    // one workgroup in 61 reports (every wave reporting serialises 32 k waves x 7 atomics on eight addresses: the kernel ran 10x slower)
    if ((tid & 63u) == 0 && (blockIdx.y * gridDim.x + blockIdx.x) % 61u == 0u) atomicAdd(&g_phase_cycles[k], now - t);
    t = now;
}
#define RAISR_PHASE_DECL unsigned long long phase_t = __builtin_amdgcn_s_memtime()
#define RAISR_PHASE(k) phase_mark(phase_t, (k), tid)
#define RAISR_PHASE_RESET phase_t = __builtin_amdgcn_s_memtime()
// the workgroup barriers of k_hashfilter_ac, timed: slot 7 collects the wave-cycles between a wave's arrival at a barrier (its own
// outstanding memory operations included: __syncthreads drains them) and its release -- "how much of a wave's life is spent parked at the
// seven barriers of a tile" (round 6; the phases keep counting their barriers too: slot 7 is "of which", not an eighth phase)
__device__ __forceinline__ void barrier_timed(unsigned tid)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63u) == 0 && (blockIdx.y * gridDim.x + blockIdx.x) % 61u == 0u) atomicAdd(&g_phase_cycles[7], t1 - t0);
}
#define RAISR_BARRIER(tid) barrier_timed(tid)
#else
#define RAISR_BARRIER(tid) __syncthreads()
#define RAISR_PHASE_DECL
#define RAISR_PHASE(k)
#define RAISR_PHASE_RESET
#endif

// XCD-aware tile order.  The dispatcher hands workgroup b to XCD b % 8 (observed, MI355X_MICROARCH.md) and
// each XCD has a private 4 MiB L2, so with the plain (blockIdx.x, blockIdx.y) order the eight tiles around
// any tile live in eight different L2s and every halo row/column is fetched from HBM again.  Remap the
// dispatch index so that each XCD walks one contiguous row-major strip of tiles: neighbouring tiles then
// share an L2 and the halo re-reads hit it.  Pure performance: any placement gives the same result.
__device__ __forceinline__ void xcd_tile(int& bx, int& by)
{
    const unsigned gx = gridDim.x, n = gridDim.x * gridDim.y;
    const unsigned b = blockIdx.y * gx + blockIdx.x;
    const unsigned n8 = n & ~7u;
    const unsigned t = b < n8 ? (b & 7u) * (n8 >> 3) + (b >> 3) : b;
    by = (int)(t / gx);
    bx = (int)(t - (unsigned)by * gx);
}

// xcd_tile for a linear tile index t of a persistent grid whose size is a multiple of 8 (so t % 8 == blockIdx.x % 8)
__device__ __forceinline__ void xcd_tile_of(unsigned t, unsigned gx, unsigned n, int& bx, int& by)
{
    const unsigned n8 = n & ~7u;
    const unsigned u = t < n8 ? (t & 7u) * (n8 >> 3) + (t >> 3) : t;
    by = (int)(u / gx);
    bx = (int)(u - (unsigned)by * gx);
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier, and on
// gfx950 the fence makes every wave wait for ALL its outstanding global loads and stores (s_waitcnt vmcnt(0)) -- which
// turns a register prefetch of the next tile into a stall, and makes a persistent workgroup wait for the write
// acknowledgement of its output stores at every tile.  The kernels below exchange nothing through global memory inside a
// workgroup, so they wait for their LDS traffic only.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// Stage an LH x LWLOAD window of a plane -- origin (y0, x0), replicate-clamped to the W x H plane --
// into an LDS tile of row stride LSTRIDE, converting each sample to TL.  Block = 256 threads, linear sweep
// (every lane busy).  All of a thread's global loads are issued before the first one is consumed: the
// rolled form waited for each load in turn, i.e. ~9 dependent memory round trips at the head of every
// workgroup.
// ------------------------------------------------------------------------------------------------
template <int LH, int LWLOAD, typename T>
struct TileRegs {
    static constexpr int REM = LWLOAD - 64;                    // halo columns right of the first 64
    static constexpr int NM = (LH + 3) / 4;                    // rows per wave in the main part
    static constexpr unsigned NR = LH * REM, NRL = (NR + 255u) / 256u;
    T vm[NM];
    T vr[NRL];
};

// global -> registers half of stage_tile (all loads in flight, nothing waits): lets a persistent workgroup fetch the
// next tile while it computes the current one
template <int LH, int LWLOAD, typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ src, int pitch, int W, int H, int y0, int x0, TileRegs<LH, LWLOAD, T>& R, unsigned tid = threadIdx.x)
{
    static_assert(LWLOAD > 64 && LWLOAD <= 128, "a tile row is one 64-lane sweep plus a remainder");
    using TR = TileRegs<LH, LWLOAD, T>;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: the row arithmetic stays scalar
    __builtin_assume(w >= 0 && w < 4);
    // main part: wave w sweeps columns [0,64) of rows w, w+4, ...: one clamped column index per lane for all rows
    const int gxm = min(max(x0 + lane, 0), W - 1);
#pragma unroll
    for (int it = 0; it < TR::NM; it++) {
        const int gy = min(max(y0 + min(w + 4 * it, LH - 1), 0), H - 1);
        R.vm[it] = src[(unsigned)gy * (unsigned)pitch + (unsigned)gxm];          // planes are < 2^31 samples: 32-bit offsets
    }
    // remainder: the REM right-hand columns of all rows, spread linearly over the block
#pragma unroll
    for (unsigned it = 0; it < TR::NRL; it++) {
        const unsigned idx = min(tid + 256u * it, TR::NR - 1u);
        const int ty = (int)(idx / TR::REM), tx = 64 + (int)(idx - (unsigned)ty * TR::REM);
        const int gy = min(max(y0 + ty, 0), H - 1), gx = min(max(x0 + tx, 0), W - 1);
        R.vr[it] = src[(unsigned)gy * (unsigned)pitch + (unsigned)gx];
    }
}

// registers -> LDS half (converting each sample to TL)
template <int LH, int LWLOAD, int LSTRIDE, typename TL, typename T>
__device__ __forceinline__ void store_tile(const TileRegs<LH, LWLOAD, T>& R, TL* sL, unsigned tid = threadIdx.x)
{
    using TR = TileRegs<LH, LWLOAD, T>;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < 4);
#pragma unroll
    for (int it = 0; it < TR::NM; it++) {
        const int ty = w + 4 * it;
        if (ty < LH) sL[ty * LSTRIDE + lane] = (TL)(float)R.vm[it];
    }
    // no `if (idx < NR)`: threads past the end loaded the LAST remainder sample (load_tile clamps idx) and store it to its own place
    // again.  With the branch the compiler sank the last sweep's load into it, behind a wait for every earlier load: two global round
    // trips in series at the head of every tile (round 5, R5.10).
#pragma unroll
    for (unsigned it = 0; it < TR::NRL; it++) {
        const unsigned idx = min(tid + 256u * it, TR::NR - 1u);
        const int ty = (int)(idx / TR::REM), tx = 64 + (int)(idx - (unsigned)ty * TR::REM);
        sL[ty * LSTRIDE + tx] = (TL)(float)R.vr[it];
    }
}

template <int LH, int LWLOAD, int LSTRIDE, typename TL, typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, int pitch, int W, int H, int y0, int x0, TL* sL, unsigned tid = threadIdx.x)
{
    TileRegs<LH, LWLOAD, T> R;
    load_tile<LH, LWLOAD>(src, pitch, W, H, y0, x0, R, tid);
    store_tile<LH, LWLOAD, LSTRIDE>(R, sL, tid);
}

