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
    printf("row_copy_pool_check: %s (%d bad)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}
