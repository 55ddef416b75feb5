// host_copy.h -- how host planes travel (included by device_abi.hip; host code only).
//
// Page-locked planes (RNLHandler_HostAlloc / raisr_hip_host_alloc / registered by the caller) go straight to the copy engines.
// PAGEABLE planes never reach a HIP copy call.  What the runtime does with pageable memory depends on shape and size (its own log,
// AMD_LOG_LEVEL=4, scripts/runtime_copy_path_probe.py): strided planes take a synchronous "unpinned rect path", small contiguous
// ones are staged, contiguous ones of a few MB are page-locked on the fly ("HSA Copy Using Pinned resource") -- the caller's memory,
// registered behind its back.  With rounds 1-2's copies the GPU suite aborted sporadically (2 of 6 first runs on a fresh box) with
// a GPU page fault inside RNLHandler_Process, always in the one test that hands a fresh strided pageable buffer per call to the
// synchronous entry while its second stream downloads row ranges; the mechanism inside the runtime was not established and the
// fault could not be provoked in isolation.  What is established: with pageable planes carried through page-locked
// BOUNCE memory the context owns -- rows packed into it (upload) or unpacked from it (download, after the copy's event) by a few
// threads -- every copy the runtime sees is a plain DMA on memory the library allocated, and the fault has not shown again
// (0 of 9 first runs).  With the last pass in row ranges the unpacking of range i overlaps the kernels of range i+1.
#pragma once
#include <immintrin.h>
#include <stdint.h>
#include <string.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- NUMA placement (the scaling risk of a multi-GPU host path is pinned memory and copy threads on the wrong socket) ------
// The bounce memory comes from hipHostMalloc with the context's device current: the runtime places it on the NUMA node closest to that
// device (no hipHostMallocNumaUser).  What the runtime cannot place are the threads that pack / unpack rows: a process-wide pool would
// copy every device's rows on whatever cores the scheduler picks.  So there is one pool PER NUMA NODE of the devices in use, its
// threads bound to that node's CPUs (within the process's own affinity mask); a context uses the pool of its device's node.
// Single-node hosts, containers without sysfs, RAISR_HIP_NUMA=0: node -1, one unbound pool, as before round 6.
// RAISR_HIP_SYSFS_ROOT: another root for /sys (tests).
namespace raisr_numa {
inline std::string sysfs_root() { const char* e = getenv("RAISR_HIP_SYSFS_ROOT"); return e ? e : "/sys"; }

// "0-15,32-47" -> CPU numbers; empty on any syntax error
inline std::vector<int> parse_cpulist(const char* text)
{
    std::vector<int> out;
    const char* p = text;
    while (*p && *p != '\n') {
        char* end = nullptr;
        const long lo = strtol(p, &end, 10);
        if (end == p || lo < 0) return {};
        long hi = lo;
        p = end;
        if (*p == '-') { hi = strtol(p + 1, &end, 10); if (end == p + 1 || hi < lo) return {}; p = end; }
        for (long c = lo; c <= hi && c < 4096; c++) out.push_back((int)c);
        if (*p == ',') p++; else if (*p && *p != '\n') return {};
    }
    return out;
}

// NUMA node of the PCI function `bdf` ("0000:c1:00.0"), -1 when unknown or the platform reports none
inline int node_of_pci(const char* bdf)
{
    if (const char* e = getenv("RAISR_HIP_NUMA")) if (e[0] == '0') return -1;
    std::string b(bdf);
    for (char& ch : b) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');
    FILE* f = fopen((sysfs_root() + "/bus/pci/devices/" + b + "/numa_node").c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

inline std::vector<int> cpus_of_node(int node)
{
    if (node < 0) return {};
    FILE* f = fopen((sysfs_root() + "/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return {};
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    return parse_cpulist(buf);
}

// does this host have more than one NUMA node with CPUs?  (one node: nothing to place)
inline bool multi_node()
{
    return !cpus_of_node(0).empty() && !cpus_of_node(1).empty();
}

// the node a context of the device with PCI address `bdf` should copy on: -1 unless the host has several nodes and the device reports one
inline int node_for_device(const char* bdf)
{
    const int node = node_of_pci(bdf);
    return (node >= 0 && multi_node() && !cpus_of_node(node).empty()) ? node : -1;
}

// bind the calling thread to the CPUs of `node` that this process may use; false (and no change) when that set is empty
inline bool bind_this_thread(int node)
{
    const std::vector<int> cpus = cpus_of_node(node);
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int n = 0;
    for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
    return n > 0 && sched_setaffinity(0, sizeof want, &want) == 0;
}
}  // namespace raisr_numa

// ---- a few persistent threads for row copies -----------------------------------------------------------------------------------
// One job at a time; its payload is cut into 128 KB blocks that the workers AND the caller claim from one atomic word
// (job number << 40 | next byte offset), so a worker that wakes up late finds nothing to claim and cannot touch a later job with
// an old descriptor; the caller returns when every claimed block has been copied (bytes-left counter), not when every worker
// has checked in, so a sleeping worker costs nothing but its share of the copy.  (Letting the workers poll between the copies of
// a frame was measured and does not pay: RAISR_HIP_COPY_SPIN.)
class RowCopyPool {
public:
    // the pool of NUMA node `node` (-1: the unbound pool); created on first use, never destroyed (its threads may outlive static destructors)
    static RowCopyPool& get(int node = -1)
    {
        static std::mutex mu;
        static std::map<int, RowCopyPool*>* pools = new std::map<int, RowCopyPool*>();
        std::lock_guard<std::mutex> lk(mu);
        RowCopyPool*& p = (*pools)[node];
        if (!p) p = new RowCopyPool(node);
        return *p;
    }
    int node() const { return node_; }
    int threads() const { return nthreads_; }
    // dst[r * dpitch .. + row_bytes) = src[r * spitch .. + row_bytes) for r < rows
    void copy(char* dst, size_t dpitch, const char* src, size_t spitch, size_t row_bytes, size_t rows)
    {
        if (!rows || !row_bytes) return;
        if (dpitch == row_bytes && spitch == row_bytes) { row_bytes *= rows; rows = 1; dpitch = spitch = row_bytes; }
        const size_t total = row_bytes * rows;
        if (nthreads_ == 0 || total < ((size_t)256 << 10)) { block(dst, dpitch, src, spitch, row_bytes, 0, total); return; }
        std::lock_guard<std::mutex> serial(serial_);         // one job at a time (contexts of several lanes share the pool)
        Job j;
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = {dst, src, dpitch, spitch, row_bytes, total, ++generation_};
            j = job_;
            left_.store(total, std::memory_order_relaxed);
            claim_.store((uint64_t)j.gen << 40, std::memory_order_release);
            gen_atomic_.store(j.gen, std::memory_order_release);
        }
        cv_.notify_all();
        work(j);
        while (left_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();      // blocks other threads are still copying
    }

private:
    struct Job { char* dst; const char* src; size_t dpitch, spitch, row_bytes, total; uint64_t gen; };
    static constexpr size_t kBlock = (size_t)128 << 10;

    explicit RowCopyPool(int node) : node_(node)
    {
        int n = 3;                                            // + the calling thread
        if (const char* e = getenv("RAISR_HIP_COPY_THREADS")) { n = atoi(e) - 1; if (n < 0) n = 0; if (n > 15) n = 15; }
        if (const char* e = getenv("RAISR_HIP_COPY_SPIN")) { spin_iters_ = atoi(e); if (spin_iters_ < 0) spin_iters_ = 0; }
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n + 1 > hw) n = (int)hw - 1;
        for (int i = 0; i < n; i++) {
            try { std::thread(&RowCopyPool::loop, this).detach(); nthreads_++; } catch (...) { break; }
        }
    }
    // memcpy with non-temporal stores for the 32-byte-aligned body: the destination of a frame copy (the caller's output plane, or the
    // bounce memory the copy engine reads next) is not read by this core again, and a 128 KB block is far below the size at which
    // glibc switches to streaming stores itself -- ordinary stores first READ every destination line (write-allocate: 3 bytes of
    // traffic per byte copied instead of 2).  Opt-in (RAISR_HIP_COPY_NT=1) until measured on the target host.
    __attribute__((target("avx2"))) static void copy_stream(char* dst, const char* src, size_t n)
    {
        const size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
        if (n < 256 + head) { memcpy(dst, src, n); return; }
        memcpy(dst, src, head);
        dst += head; src += head; n -= head;
        const size_t body = n & ~(size_t)127;
        for (size_t i = 0; i < body; i += 128) {
            const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
            const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
            _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
            _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
        }
        memcpy(dst + body, src + body, n - body);
    }
    static bool use_stream_stores()
    {
        static const bool on = __builtin_cpu_supports("avx2") && getenv("RAISR_HIP_COPY_NT") && atoi(getenv("RAISR_HIP_COPY_NT")) == 1;
        return on;
    }
    // byte range [from, to) of the payload (row-major over rows x row_bytes)
    static void block(char* dst, size_t dpitch, const char* src, size_t spitch, size_t row_bytes, size_t from, size_t to)
    {
        const bool nt = use_stream_stores();
        while (from < to) {
            const size_t r = from / row_bytes, x = from % row_bytes;
            size_t n = row_bytes - x;
            if (n > to - from) n = to - from;
            if (nt) copy_stream(dst + r * dpitch + x, src + r * spitch + x, n);
            else memcpy(dst + r * dpitch + x, src + r * spitch + x, n);
            from += n;
        }
        if (nt) _mm_sfence();                                // streaming stores are weakly ordered: fence before the block is reported done
    }
    void work(const Job& j)
    {
        for (;;) {
            uint64_t cur = claim_.load(std::memory_order_acquire);
            size_t from;
            for (;;) {
                if ((cur >> 40) != (j.gen & 0xFFFFFFu)) return;           // a later job owns the word: nothing of this one is left
                from = (size_t)(cur & (((uint64_t)1 << 40) - 1));
                if (from >= j.total) return;
                if (claim_.compare_exchange_weak(cur, cur + kBlock, std::memory_order_acq_rel)) break;
            }
            const size_t to = from + kBlock < j.total ? from + kBlock : j.total;
            block(j.dst, j.dpitch, j.src, j.spitch, j.row_bytes, from, to);
            left_.fetch_sub(to - from, std::memory_order_acq_rel);
        }
    }
    void loop()
    {
        if (node_ >= 0) (void)raisr_numa::bind_this_thread(node_);     // the rows this pool copies live on node_'s memory
        uint64_t seen = 0;
        for (;;) {
            bool have = false;
            for (int spin = 0; spin < spin_iters_ && !have; spin++) {
                if (gen_atomic_.load(std::memory_order_acquire) != seen) have = true;
                else __builtin_ia32_pause();
            }
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!have) cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                j = job_;
            }
            work(j);
        }
    }

    std::mutex serial_, mu_;
    std::condition_variable cv_;
    Job job_{};
    uint64_t generation_ = 0;
    std::atomic<uint64_t> gen_atomic_{0}, claim_{0};
    std::atomic<size_t> left_{0};
    int spin_iters_ = 0;                                     // RAISR_HIP_COPY_SPIN=n: poll n times for the next job before sleeping (measured: no gain, 3 busy cores)
    int nthreads_ = 0;
    const int node_;
};

// ---- per-context bounce memory -------------------------------------------------------------------------------------------------
struct HostBounce {
    int node = -1;                           // NUMA node of the owning context's device: which RowCopyPool unpacks (raisr_numa)
    char* base = nullptr;
    size_t bytes = 0, used = 0, want = 0;
    struct Unpack { char* dst; size_t dpitch; const char* src; size_t row_bytes, rows; hipEvent_t ev; };
    std::vector<Unpack> unpack;              // downloads whose bytes still sit in the bounce memory
    std::vector<hipEvent_t> events;          // one per bounced copy of a frame (uploads and downloads), re-used frame after frame
    size_t ev_used = 0;
    std::vector<hipEvent_t> uploads;         // events behind this frame's bounced uploads: the DMA engine may still be reading the memory

    // first and last byte in the runtime's table of page-locked memory
    static bool page_locked(const void* p, size_t n)
    {
        return n && raisr_hip_host_is_page_locked(p) && raisr_hip_host_is_page_locked((const char*)p + n - 1);
    }
    // a new frame starts: it may need up to `need` bytes.  The previous frame's downloads have been unpacked by the caller
    // (finish()); its UPLOADS are waited for here -- a second asynchronous frame on the same context without a synchronise in
    // between would otherwise repack rows into memory the copy engine is still reading (or free it in take()).
    // The memory itself is allocated by the first plane that turns out to be pageable.
    void begin_frame(size_t need)
    {
        for (hipEvent_t e : uploads) (void)hipEventSynchronize(e);
        uploads.clear();
        used = 0;
        ev_used = 0;
        want = need;
    }
    char* take(size_t n)
    {
        if (bytes < want) {
            if (used) return nullptr;                    // cannot grow under copies in flight (want is fixed per frame: not reached)
            if (base) (void)hipHostFree(base);
            base = nullptr; bytes = 0;
            if (hipHostMalloc((void**)&base, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); base = nullptr; return nullptr; }
            bytes = want;
        }
        const size_t at = (used + 255) & ~(size_t)255;
        if (!base || at + n > bytes) return nullptr;
        used = at + n;
        return base + at;
    }
    hipEvent_t next_event()
    {
        if (ev_used == events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            events.push_back(e);
        }
        return events[ev_used++];
    }
    // every download that went through the bounce memory: wait for its copy, then put the rows where the caller wants them
    hipError_t finish()
    {
        hipError_t first = hipSuccess;
        for (const Unpack& u : unpack) {
            const hipError_t e = hipEventSynchronize(u.ev);
            if (e != hipSuccess) { if (first == hipSuccess) first = e; continue; }
            RowCopyPool::get(node).copy(u.dst, u.dpitch, u.src, u.row_bytes, u.row_bytes, u.rows);
        }
        unpack.clear();
        return first;
    }
    void release()
    {
        for (hipEvent_t e : uploads) (void)hipEventSynchronize(e);
        uploads.clear();
        unpack.clear();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        if (base) (void)hipHostFree(base);
        base = nullptr; bytes = used = want = 0;
    }
};
