// host_sha256.h — SHA-256 for the Fiat–Shamir transcript of `helpers::compute_challenge`
// (primitives/src/helpers.rs:411-472; the reference uses the sha2 0.10 crate).  A single hash stream is sequential, so it
// runs on a host core (x86 SHA extensions when the CPU has them, portable code otherwise) BESIDE the GPU work of the same call.
// Incremental interface: the transcript prefix (tag, length, evaluations) can be absorbed before the commitment is known.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzg_host {

struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buf_len;
    uint64_t total;
};

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void sha256_blocks_portable(uint32_t h[8], const uint8_t* p, size_t n_blocks) {
    for (; n_blocks; --n_blocks, p += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + SHA256_K[i] + w[i];
            uint32_t S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
}

#if defined(__x86_64__)
// x86 SHA extensions (sha256rnds2 / sha256msg1 / sha256msg2): two rounds per instruction
__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_shani(uint32_t h[8], const uint8_t* p, size_t n_blocks) {
    const __m128i shuf = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&h[0]);         // DCBA
    __m128i st1 = _mm_loadu_si128((const __m128i*)&h[4]);         // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                           // CDAB
    st1 = _mm_shuffle_epi32(st1, 0x1B);                           // EFGH
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                   // ABEF
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);                        // CDGH
    for (; n_blocks; --n_blocks, p += 64) {
        const __m128i save0 = st0, save1 = st1;
        __m128i m0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), shuf);
        __m128i m1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), shuf);
        __m128i m2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), shuf);
        __m128i m3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), shuf);
        __m128i msg;
#define KZG_SHA_RND(mi, kidx)                                                              \
        msg = _mm_add_epi32(mi, _mm_loadu_si128((const __m128i*)&SHA256_K[kidx]));         \
        st1 = _mm_sha256rnds2_epu32(st1, st0, msg);                                        \
        msg = _mm_shuffle_epi32(msg, 0x0E);                                                \
        st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
        KZG_SHA_RND(m0, 0)
        KZG_SHA_RND(m1, 4)
        KZG_SHA_RND(m2, 8)
        KZG_SHA_RND(m3, 12)
        // message schedule, four words at a time: with a, b, c, d = the last four groups (a oldest),
        // next = msg2(msg1(a, b) + alignr(d, c, 4), d); it replaces a
#define KZG_SHA_NEXT(a, b, c, d) a = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(a, b), _mm_alignr_epi8(d, c, 4)), d);
        KZG_SHA_NEXT(m0, m1, m2, m3) KZG_SHA_RND(m0, 16)
        KZG_SHA_NEXT(m1, m2, m3, m0) KZG_SHA_RND(m1, 20)
        KZG_SHA_NEXT(m2, m3, m0, m1) KZG_SHA_RND(m2, 24)
        KZG_SHA_NEXT(m3, m0, m1, m2) KZG_SHA_RND(m3, 28)
        KZG_SHA_NEXT(m0, m1, m2, m3) KZG_SHA_RND(m0, 32)
        KZG_SHA_NEXT(m1, m2, m3, m0) KZG_SHA_RND(m1, 36)
        KZG_SHA_NEXT(m2, m3, m0, m1) KZG_SHA_RND(m2, 40)
        KZG_SHA_NEXT(m3, m0, m1, m2) KZG_SHA_RND(m3, 44)
        KZG_SHA_NEXT(m0, m1, m2, m3) KZG_SHA_RND(m0, 48)
        KZG_SHA_NEXT(m1, m2, m3, m0) KZG_SHA_RND(m1, 52)
        KZG_SHA_NEXT(m2, m3, m0, m1) KZG_SHA_RND(m2, 56)
        KZG_SHA_NEXT(m3, m0, m1, m2) KZG_SHA_RND(m3, 60)
#undef KZG_SHA_NEXT
#undef KZG_SHA_RND
        st0 = _mm_add_epi32(st0, save0);
        st1 = _mm_add_epi32(st1, save1);
    }
    tmp = _mm_shuffle_epi32(st0, 0x1B);                           // FEBA
    st1 = _mm_shuffle_epi32(st1, 0xB1);                           // DCHG
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);                        // DCBA
    st1 = _mm_alignr_epi8(st1, tmp, 8);                           // HGFE
    _mm_storeu_si128((__m128i*)&h[0], st0);
    _mm_storeu_si128((__m128i*)&h[4], st1);
}
// TWO independent messages, n_blocks each, interleaved in one instruction stream: a single stream is bound by the latency of its sha256rnds2
// chain (two dependent instructions per four rounds), a second message fills the empty issue slots (tools/ubench/sha_x2.cpp measures the ratio).
__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_shani_x2(uint32_t ha[8], const uint8_t* pa, uint32_t hb[8], const uint8_t* pb, size_t n_blocks) {
    const __m128i shuf = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i a0, a1, b0, b1;
    {
        __m128i tmp = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&ha[0]), 0xB1), s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&ha[4]), 0x1B);
        a0 = _mm_alignr_epi8(tmp, s1, 8); a1 = _mm_blend_epi16(s1, tmp, 0xF0);
        tmp = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&hb[0]), 0xB1); s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)&hb[4]), 0x1B);
        b0 = _mm_alignr_epi8(tmp, s1, 8); b1 = _mm_blend_epi16(s1, tmp, 0xF0);
    }
    for (; n_blocks; --n_blocks, pa += 64, pb += 64) {
        const __m128i sa0 = a0, sa1 = a1, sb0 = b0, sb1 = b1;
        __m128i x0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 0)), shuf), x1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 16)), shuf);
        __m128i x2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 32)), shuf), x3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 48)), shuf);
        __m128i y0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 0)), shuf), y1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 16)), shuf);
        __m128i y2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 32)), shuf), y3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 48)), shuf);
        __m128i ma, mb, kk;
#define KZG_SHA_RND2(xi, yi, kidx)                                                         \
        kk = _mm_loadu_si128((const __m128i*)&SHA256_K[kidx]);                             \
        ma = _mm_add_epi32(xi, kk); mb = _mm_add_epi32(yi, kk);                            \
        a1 = _mm_sha256rnds2_epu32(a1, a0, ma); b1 = _mm_sha256rnds2_epu32(b1, b0, mb);    \
        ma = _mm_shuffle_epi32(ma, 0x0E); mb = _mm_shuffle_epi32(mb, 0x0E);                \
        a0 = _mm_sha256rnds2_epu32(a0, a1, ma); b0 = _mm_sha256rnds2_epu32(b0, b1, mb);
#define KZG_SHA_NEXT2(a, b, c, d) a = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(a, b), _mm_alignr_epi8(d, c, 4)), d);
        KZG_SHA_RND2(x0, y0, 0) KZG_SHA_RND2(x1, y1, 4) KZG_SHA_RND2(x2, y2, 8) KZG_SHA_RND2(x3, y3, 12)
        KZG_SHA_NEXT2(x0, x1, x2, x3) KZG_SHA_NEXT2(y0, y1, y2, y3) KZG_SHA_RND2(x0, y0, 16)
        KZG_SHA_NEXT2(x1, x2, x3, x0) KZG_SHA_NEXT2(y1, y2, y3, y0) KZG_SHA_RND2(x1, y1, 20)
        KZG_SHA_NEXT2(x2, x3, x0, x1) KZG_SHA_NEXT2(y2, y3, y0, y1) KZG_SHA_RND2(x2, y2, 24)
        KZG_SHA_NEXT2(x3, x0, x1, x2) KZG_SHA_NEXT2(y3, y0, y1, y2) KZG_SHA_RND2(x3, y3, 28)
        KZG_SHA_NEXT2(x0, x1, x2, x3) KZG_SHA_NEXT2(y0, y1, y2, y3) KZG_SHA_RND2(x0, y0, 32)
        KZG_SHA_NEXT2(x1, x2, x3, x0) KZG_SHA_NEXT2(y1, y2, y3, y0) KZG_SHA_RND2(x1, y1, 36)
        KZG_SHA_NEXT2(x2, x3, x0, x1) KZG_SHA_NEXT2(y2, y3, y0, y1) KZG_SHA_RND2(x2, y2, 40)
        KZG_SHA_NEXT2(x3, x0, x1, x2) KZG_SHA_NEXT2(y3, y0, y1, y2) KZG_SHA_RND2(x3, y3, 44)
        KZG_SHA_NEXT2(x0, x1, x2, x3) KZG_SHA_NEXT2(y0, y1, y2, y3) KZG_SHA_RND2(x0, y0, 48)
        KZG_SHA_NEXT2(x1, x2, x3, x0) KZG_SHA_NEXT2(y1, y2, y3, y0) KZG_SHA_RND2(x1, y1, 52)
        KZG_SHA_NEXT2(x2, x3, x0, x1) KZG_SHA_NEXT2(y2, y3, y0, y1) KZG_SHA_RND2(x2, y2, 56)
        KZG_SHA_NEXT2(x3, x0, x1, x2) KZG_SHA_NEXT2(y3, y0, y1, y2) KZG_SHA_RND2(x3, y3, 60)
#undef KZG_SHA_NEXT2
#undef KZG_SHA_RND2
        a0 = _mm_add_epi32(a0, sa0); a1 = _mm_add_epi32(a1, sa1);
        b0 = _mm_add_epi32(b0, sb0); b1 = _mm_add_epi32(b1, sb1);
    }
    {
        __m128i tmp = _mm_shuffle_epi32(a0, 0x1B), s1 = _mm_shuffle_epi32(a1, 0xB1);
        _mm_storeu_si128((__m128i*)&ha[0], _mm_blend_epi16(tmp, s1, 0xF0)); _mm_storeu_si128((__m128i*)&ha[4], _mm_alignr_epi8(s1, tmp, 8));
        tmp = _mm_shuffle_epi32(b0, 0x1B); s1 = _mm_shuffle_epi32(b1, 0xB1);
        _mm_storeu_si128((__m128i*)&hb[0], _mm_blend_epi16(tmp, s1, 0xF0)); _mm_storeu_si128((__m128i*)&hb[4], _mm_alignr_epi8(s1, tmp, 8));
    }
}
inline bool sha256_have_shani() {
    static const bool have = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return have;
}
#endif

inline void sha256_blocks(uint32_t h[8], const uint8_t* p, size_t n_blocks) {
#if defined(__x86_64__)
    if (sha256_have_shani()) { sha256_blocks_shani(h, p, n_blocks); return; }
#endif
    sha256_blocks_portable(h, p, n_blocks);
}

inline void sha256_init(Sha256& s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s.h, iv, sizeof iv);
    s.buf_len = 0;
    s.total = 0;
}
inline void sha256_update(Sha256& s, const uint8_t* data, size_t len) {
    s.total += len;
    if (s.buf_len) {
        size_t take = 64 - s.buf_len < len ? 64 - s.buf_len : len;
        memcpy(s.buf + s.buf_len, data, take);
        s.buf_len += take; data += take; len -= take;
        if (s.buf_len == 64) { sha256_blocks(s.h, s.buf, 1); s.buf_len = 0; }
    }
    if (len >= 64) {
        size_t nb = len / 64;
        sha256_blocks(s.h, data, nb);
        data += nb * 64; len -= nb * 64;
    }
    if (len) { memcpy(s.buf, data, len); s.buf_len = len; }
}
inline void sha256_final(Sha256& s, uint8_t out[32]) {
    uint8_t pad[128];
    size_t n = 0;
    pad[n++] = 0x80;
    while ((s.buf_len + n) % 64 != 56) pad[n++] = 0;
    const uint64_t bits = s.total * 8;
    for (int i = 7; i >= 0; --i) pad[n++] = (uint8_t)(bits >> (8 * i));
    const uint64_t keep = s.total;
    sha256_update(s, pad, n);
    s.total = keep;
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(s.h[i] >> 24); out[4 * i + 1] = (uint8_t)(s.h[i] >> 16); out[4 * i + 2] = (uint8_t)(s.h[i] >> 8); out[4 * i + 3] = (uint8_t)s.h[i]; }
}

}  // namespace kzg_host
