// tests/hostcheck/transcriptcheck.cpp — TEST-ONLY host build of csrc/host_transcript.h (the transcript-prefix segment generator and SHA-256 over two
// streams at once), loaded by tests/test_challenge_host.py.  Not part of the product library.
#include <cstdint>
#include "host_transcript.h"

using namespace kzg_host;

extern "C" {
// SHA-256 of the transcript prefix of blob a alone (out_a1), and of a and b absorbed TOGETHER through sha256_absorb_x2 (out_a2, out_b2)
void tc_prefix_digests(const uint8_t* a, size_t len_a, size_t np_a, const uint8_t* b, size_t len_b, size_t np_b, uint8_t out_a1[32], uint8_t out_a2[32], uint8_t out_b2[32]) {
    Sha256 s1; sha256_init(s1);
    TranscriptPrefix g1(a, len_a, np_a);
    sha256_absorb(s1, g1);
    sha256_final(s1, out_a1);
    Sha256 sa, sb; sha256_init(sa); sha256_init(sb);
    TranscriptPrefix ga(a, len_a, np_a), gb(b, len_b, np_b);
    sha256_absorb_x2(sa, ga, sb, gb);
    sha256_final(sa, out_a2);
    sha256_final(sb, out_b2);
}
size_t tc_prefix_blocks(size_t n_padded) { return TranscriptPrefix::blocks(n_padded); }
int tc_have_shani(void) {
#if defined(__x86_64__)
    return sha256_have_shani() ? 1 : 0;
#else
    return 0;
#endif
}
}

#ifdef TRANSCRIPT_SELF_CHECK
// self-check under the sanitizers (tests/test_sanitizers_host.py): the two-stream form against the one-stream form over pseudo-random pairs of lengths,
// canonical and non-canonical chunks (buffers sized exactly: an over-read of a segment trips ASAN)
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    uint64_t x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    int cases = 0;
    for (int it = 0; it < 300; ++it) {
        const size_t la = 1 + rnd() % (it % 7 == 0 ? 70000 : 3000), lb = it % 5 == 0 ? la : 1 + rnd() % 5000;
        std::vector<uint8_t> a(la), b(lb);
        for (auto& v : a) v = (uint8_t)rnd();
        for (auto& v : b) v = (uint8_t)rnd();
        if (it & 1) for (size_t i = 0; i < la; i += 32) a[i] &= 0x1F;
        if (it & 2) for (size_t i = 0; i < lb; i += 32) b[i] &= 0x1F;
        auto np = [](size_t len) { size_t e = (len + 31) / 32, p = 1; while (p < e) p <<= 1; return p; };
        uint8_t a1[32], a2[32], b2[32], b1[32], t1[32], t2[32];
        tc_prefix_digests(a.data(), la, np(la), b.data(), lb, np(lb), a1, a2, b2);
        tc_prefix_digests(b.data(), lb, np(lb), a.data(), la, np(la), b1, t1, t2);
        if (memcmp(a1, a2, 32) || memcmp(b1, b2, 32) || memcmp(t1, b1, 32) || memcmp(t2, a1, 32)) { printf("MISMATCH at case %d (%zu, %zu)\n", it, la, lb); return 1; }
        ++cases;
    }
    printf("transcript self-check ok: %d pairs, SHA extensions %d\n", cases, tc_have_shani());
    return 0;
}
#endif
