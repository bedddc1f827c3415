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
