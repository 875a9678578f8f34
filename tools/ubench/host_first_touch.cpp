// What does the first touch of a fresh 160 MB host buffer cost (configs[4]'s scores: np.empty((1000000, 20)) filled by plspm_fit), and does it scale with
// the number of threads that touch it?  T threads copy 32 MB chunks in stripes into a fresh anonymous mapping (mode 0), one advised MADV_HUGEPAGE as NumPy
// advises its large allocations (mode 1), or one populated first by T threads calling madvise(MADV_POPULATE_WRITE) on their own stripes (mode 2: the time
// includes the populate).  Build: g++ -O2 -pthread tools/ubench/host_first_touch.cpp -o /tmp/host_first_touch
#include <sys/mman.h>
#include <cstring>
#include <cstdio>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdlib>
#include <algorithm>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
int main(int argc, char** argv) {
    const size_t bytes = (size_t)160 << 20, half = (size_t)32 << 20;
    char* src = (char*)malloc(half); memset(src, 1, half);
    for (int mode = 0; mode < 3; ++mode)
        for (int T : {1, 4, 8, 16, 32}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                char* dst = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (mode >= 1) madvise(dst, bytes, MADV_HUGEPAGE);
                auto t0 = std::chrono::steady_clock::now();
                if (mode == 2) {
                    std::vector<std::thread> th;
                    const size_t per = ((bytes / T) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
                    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1)); if (hi > lo) madvise(dst + lo, hi - lo, MADV_POPULATE_WRITE); });
                    for (auto& x : th) x.join();
                }
                for (size_t off = 0; off < bytes; off += half) {
                    const size_t n = std::min(half, bytes - off);
                    std::vector<std::thread> th;
                    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { const size_t per = ((n / T) + 4095) & ~(size_t)4095, lo = std::min(n, per * t), hi = (t == T - 1) ? n : std::min(n, per * (t + 1)); if (hi > lo) memcpy(dst + off + lo, src + lo, hi - lo); });
                    for (auto& x : th) x.join();
                }
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                munmap(dst, bytes);
            }
            printf("{\"mode\": %d, \"threads\": %d, \"ms\": %.2f, \"GBps\": %.1f}\n", mode, T, best, bytes / best / 1e6);
        }
    return 0;
}
