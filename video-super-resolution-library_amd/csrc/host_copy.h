// host_copy.h -- how host planes travel (included by device_abi.hip; host code only).
//
// Page-locked planes (RNLHandler_HostAlloc / raisr_hip_host_alloc / registered by the caller) go straight to the copy engines.
// PAGEABLE planes never reach an asynchronous HIP copy: the runtime serves such a copy by page-locking the caller's memory on
// the fly and REMEMBERING the registration past the call; when the host then frees the buffer and its allocator hands the
// address out again (a frame buffer per call is ordinary host behaviour), the next copy goes through the remembered mapping,
// which the driver may have dropped in the meantime -- a GPU page fault that aborts the process (seen in 2 of 10 runs of the
// GPU suite with one test that passes a fresh strided array per call).  So pageable planes are carried through page-locked
// BOUNCE memory the context owns: rows are packed into it (upload) or unpacked from it (download, after the copy's event) by a
// few threads; with the last pass in row ranges the unpacking of range i overlaps the kernels of range i+1.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// ---- a few persistent threads for row copies -----------------------------------------------------------------------------------
class RowCopyPool {
public:
    static RowCopyPool& get()
    {
        static RowCopyPool* p = new RowCopyPool();          // never destroyed: its threads may outlive static destructors
        return *p;
    }
    // dst[r * dpitch .. + row_bytes) = src[r * spitch .. + row_bytes) for r < rows, cut into blocks the threads (and the caller) take
    void copy(char* dst, size_t dpitch, const char* src, size_t spitch, size_t row_bytes, size_t rows)
    {
        if (!rows || !row_bytes) return;
        if (dpitch == row_bytes && spitch == row_bytes) { row_bytes *= rows; rows = 1; dpitch = spitch = row_bytes; }
        const size_t total = row_bytes * rows;
        if (nthreads_ == 0 || total < ((size_t)256 << 10)) { block(dst, dpitch, src, spitch, row_bytes, rows, 0, total); return; }
        std::unique_lock<std::mutex> serial(serial_);        // one job at a time (contexts of several lanes share the pool)
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = {dst, src, dpitch, spitch, row_bytes, rows, total};
            next_.store(0, std::memory_order_relaxed);
            pending_ = nthreads_;
            generation_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    struct Job { char* dst; const char* src; size_t dpitch, spitch, row_bytes, rows, total; };
    static constexpr size_t kBlock = (size_t)128 << 10;

    RowCopyPool()
    {
        int n = 3;                                            // + the calling thread
        if (const char* e = getenv("RAISR_HIP_COPY_THREADS")) { n = atoi(e) - 1; if (n < 0) n = 0; if (n > 15) n = 15; }
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n + 1 > hw) n = (int)hw - 1;
        for (int i = 0; i < n; i++) {
            try { std::thread(&RowCopyPool::loop, this).detach(); nthreads_++; } catch (...) { break; }
        }
    }
    // byte range [from, to) of the job's payload (row-major over rows x row_bytes)
    static void block(char* dst, size_t dpitch, const char* src, size_t spitch, size_t row_bytes, size_t rows, size_t from, size_t to)
    {
        (void)rows;
        while (from < to) {
            const size_t r = from / row_bytes, x = from % row_bytes;
            size_t n = row_bytes - x;
            if (n > to - from) n = to - from;
            memcpy(dst + r * dpitch + x, src + r * spitch + x, n);
            from += n;
        }
    }
    void work()
    {
        const Job j = job_;
        for (;;) {
            const size_t from = next_.fetch_add(kBlock, std::memory_order_relaxed);
            if (from >= j.total) break;
            block(j.dst, j.dpitch, j.src, j.spitch, j.row_bytes, j.rows, from, from + kBlock < j.total ? from + kBlock : j.total);
        }
    }
    void loop()
    {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            work();
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }

    std::mutex serial_, mu_;
    std::condition_variable cv_, done_;
    Job job_{};
    std::atomic<size_t> next_{0};
    unsigned generation_ = 0;
    int pending_ = 0;
    int nthreads_ = 0;
};

// ---- per-context bounce memory -------------------------------------------------------------------------------------------------
struct HostBounce {
    char* base = nullptr;
    size_t bytes = 0, used = 0, want = 0;
    struct Unpack { char* dst; size_t dpitch; const char* src; size_t row_bytes, rows; hipEvent_t ev; };
    std::vector<Unpack> unpack;              // downloads whose bytes still sit in the bounce memory
    std::vector<hipEvent_t> events;          // one per download of a frame, re-used frame after frame
    size_t ev_used = 0;

    // first and last byte in the runtime's table of page-locked memory
    static bool page_locked(const void* p, size_t n)
    {
        return n && raisr_hip_host_is_page_locked(p) && raisr_hip_host_is_page_locked((const char*)p + n - 1);
    }
    // a new frame starts: it may need up to `need` bytes; nothing of the previous frame may be pending (the caller synchronised).
    // The memory itself is allocated by the first plane that turns out to be pageable.
    void begin_frame(size_t need)
    {
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
            RowCopyPool::get().copy(u.dst, u.dpitch, u.src, u.row_bytes, u.row_bytes, u.rows);
        }
        unpack.clear();
        return first;
    }
    void release()
    {
        unpack.clear();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        if (base) (void)hipHostFree(base);
        base = nullptr; bytes = used = want = 0;
    }
};
