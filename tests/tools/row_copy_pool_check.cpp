// CPU check of csrc/host_copy.h's RowCopyPool (the threads that pack / unpack pageable host planes): random strided copies from
// several caller threads at once, byte-compared with a plain loop; bytes outside the rows must stay untouched.
// Built by tests/test_host_copy_pool.py:  g++ -std=c++17 -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude ...
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "raisr_hip.h"
#include "../../video-super-resolution-library_amd/csrc/host_copy.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static int one_case(uint64_t seed)
{
    uint64_t st = seed * 2654435761u + 12345;
    auto r = [&](uint64_t m) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st % m; };
    const size_t row_bytes = 1 + r(5000), rows = 1 + r(700);
    const size_t spitch = r(3) ? row_bytes + r(97) : row_bytes, dpitch = r(3) ? row_bytes + r(131) : row_bytes;
    std::vector<uint8_t> src(spitch * rows + 64), dst(dpitch * rows + 64, 0xA5), want(dpitch * rows + 64, 0xA5);
    for (auto& b : src) b = (uint8_t)r(256);
    for (size_t y = 0; y < rows; y++) memcpy(&want[y * dpitch], &src[y * spitch], row_bytes);
    RowCopyPool::get().copy((char*)dst.data(), dpitch, (const char*)src.data(), spitch, row_bytes, rows);
    return memcmp(dst.data(), want.data(), dst.size()) == 0 ? 0 : 1;
}

int main()
{
    int bad = 0;
    for (int i = 0; i < 60; i++) bad += one_case(rnd());
    {   // a large contiguous copy (the whole-plane case)
        const size_t n = (size_t)9 << 20;
        std::vector<uint8_t> a(n), b(n, 0);
        for (size_t i = 0; i < n; i += 97) a[i] = (uint8_t)(i * 31);
        RowCopyPool::get().copy((char*)b.data(), n / 2160, (const char*)a.data(), n / 2160, n / 2160, 2160);
        bad += memcmp(a.data(), b.data(), (n / 2160) * 2160) != 0;
    }
    std::vector<std::thread> callers;                    // lanes of a ring share the pool
    std::atomic<int> bad_mt{0};
    for (int t = 0; t < 4; t++)
        callers.emplace_back([&, t] { for (int i = 0; i < 40; i++) bad_mt += one_case(1000003ull * (unsigned)t + (unsigned)i); });
    for (auto& th : callers) th.join();
    bad += bad_mt.load();
    // NUMA placement (raisr_numa): with RAISR_HIP_SYSFS_ROOT pointing at a fake two-node tree (tests/test_host_copy_pool.py builds it from
    // this process's own CPUs), a device on node 1 gets the node-1 pool, whose threads run on node 1's CPUs only and still copy correctly
    if (getenv("RAISR_HIP_SYSFS_ROOT")) {
        using namespace raisr_numa;
        bad += parse_cpulist("0-2,5,7-8\n") != std::vector<int>({0, 1, 2, 5, 7, 8});
        bad += !parse_cpulist("3-1").empty() || !parse_cpulist("a").empty() || !parse_cpulist("1;2").empty();
        bad += node_of_pci("0000:C1:00.0") != 1 || node_of_pci("0000:01:00.0") != 0 || node_of_pci("0000:ff:00.0") != -1;
        bad += !multi_node() || node_for_device("0000:c1:00.0") != 1;
        RowCopyPool& p1 = RowCopyPool::get(1);
        bad += p1.node() != 1 || &p1 == &RowCopyPool::get(-1) || &p1 != &RowCopyPool::get(1);
        const size_t n = (size_t)6 << 20;
        std::vector<uint8_t> a(n), b(n, 0);
        for (size_t i = 0; i < n; i += 61) a[i] = (uint8_t)(i * 17);
        for (int rep = 0; rep < 8; rep++) p1.copy((char*)b.data(), n / 1080, (const char*)a.data(), n / 1080, n / 1080, 1080);
        bad += memcmp(a.data(), b.data(), (n / 1080) * 1080) != 0;
        // the pool's threads are bound: ask one of them through a copy whose "memcpy" we cannot hook -- instead bind THIS thread the same way
        // and compare with node 1's list
        cpu_set_t before; sched_getaffinity(0, sizeof before, &before);
        if (bind_this_thread(1)) {
            cpu_set_t now; sched_getaffinity(0, sizeof now, &now);
            for (int c : cpus_of_node(0)) { bool in1 = false; for (int d : cpus_of_node(1)) in1 |= d == c; if (!in1 && CPU_ISSET(c, &now)) bad++; }
            sched_setaffinity(0, sizeof before, &before);
        } else bad++;
        setenv("RAISR_HIP_NUMA", "0", 1);
        bad += node_for_device("0000:c1:00.0") != -1;
        unsetenv("RAISR_HIP_NUMA");
    }
    printf("row_copy_pool_check: %s (%d bad)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}
