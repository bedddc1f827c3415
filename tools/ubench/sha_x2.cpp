// Host micro-benchmark: ONE SHA-256 stream through the x86 SHA extensions against TWO independent streams interleaved in one thread.
// A single stream is bound by the latency of its sha256rnds2 chain (two dependent instructions per four rounds); a second, independent message
// fills the empty issue slots.  Build: g++ -O3 -std=c++17 -Irust-kzg-bn254_amd/csrc tools/ubench/sha_x2.cpp -o tools/ubench/sha_x2
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "host_sha256.h"
using namespace kzg_host;
int main(int argc, char** argv) {
    const size_t len = (argc > 1 ? atol(argv[1]) : 25000) / 64 * 64, reps = argc > 2 ? atol(argv[2]) : 4000;
    std::vector<uint8_t> a(len), b(len);
    for (size_t i = 0; i < len; ++i) { a[i] = (uint8_t)(i * 131 + 7); b[i] = (uint8_t)(i * 197 + 3); }
    uint32_t h1[8], h2[8], g1[8], g2[8];
    Sha256 s; sha256_init(s);
    auto t0 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < reps; ++r) {
        memcpy(h1, s.h, 32); memcpy(h2, s.h, 32);
        sha256_blocks_shani(h1, a.data(), len / 64);
        sha256_blocks_shani(h2, b.data(), len / 64);
    }
    auto t1 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < reps; ++r) {
        memcpy(g1, s.h, 32); memcpy(g2, s.h, 32);
        sha256_blocks_shani_x2(g1, a.data(), g2, b.data(), len / 64);
    }
    auto t2 = std::chrono::steady_clock::now();
    const double one = std::chrono::duration<double>(t1 - t0).count(), two = std::chrono::duration<double>(t2 - t1).count();
    const double bytes = 2.0 * len * reps;
    printf("%zu-byte messages: one stream at a time %.2f GB/s, two interleaved %.2f GB/s (x %.2f); digests %s\n", len, bytes / one / 1e9, bytes / two / 1e9, one / two,
           memcmp(h1, g1, 32) == 0 && memcmp(h2, g2, 32) == 0 ? "equal" : "DIFFER");
    return memcmp(h1, g1, 32) || memcmp(h2, g2, 32);
}
