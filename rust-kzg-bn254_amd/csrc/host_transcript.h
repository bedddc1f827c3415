// host_transcript.h — the byte stream of helpers::compute_challenge's transcript prefix (primitives/src/helpers.rs:411-455) as a sequence of
// segments, and SHA-256 over TWO such streams at once.  Host only (also compiled with g++ by tests/hostcheck).
//
//   prefix = "EIGENDA_FSBLOBVERIFY_V1_" || u64be(n) || n x 32 bytes: every 32-byte big-endian chunk of the blob reduced mod r (Blob::to_polynomial_eval_form ->
//            to_byte_array, helpers.rs:40-57, :80-119), the last chunk right-padded with zeros, zero elements up to the next power of two
// Canonical chunks (the normal case) are hashed straight from the caller's buffer: the generator hands out runs of them; a chunk >= r ends the
// run and is handed out reduced, from the generator's own 32 bytes.
//
// Two streams: one SHA-256 stream through the x86 SHA extensions is bound by the latency of its sha256rnds2 chain; the transcripts of two blobs of a
// batch are independent, and interleaved in one thread they run at 1.5 x the single-stream rate (EPYC 9575F: 2.44 -> 3.65 GB/s, tools/ubench/sha_x2.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include "host_pairing.h"
#include "host_sha256.h"

namespace kzg_host {

struct ShaSeg { const uint8_t* p; size_t len; };

class TranscriptPrefix {
public:
    TranscriptPrefix(const uint8_t* blob, size_t len, size_t n_padded) : blob_(blob), len_(len), n_full_(len / 32), n_padded_(n_padded) {
        memcpy(hdr_, "EIGENDA_FSBLOBVERIFY_V1_", 24);                                   // primitives/src/consts.rs:8
        for (int i = 0; i < 8; ++i) hdr_[24 + i] = (uint8_t)((uint64_t)n_padded >> (8 * (7 - i)));
        for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) r_be_[8 * i + b] = (uint8_t)(FR_MODULUS_WORDS[3 - i] >> (8 * (7 - b)));
    }
    // the next segment; false when the prefix is exhausted
    bool operator()(ShaSeg& s) {
        for (;;) {
            switch (stage_) {
                case 0: stage_ = 1; s = ShaSeg{hdr_, 32}; return true;
                case 1: {                                                                 // whole chunks: runs of canonical ones, reduced ones singly
                    if (pending_reduced_) { pending_reduced_ = false; s = ShaSeg{red_, 32}; return true; }
                    // (a run is handed out after RUN_MAX chunks at the latest: the hash then reads them from the cache the scan left them in, not from memory again)
                    size_t i = run_;
                    const size_t first = run_, stop = n_full_ - first > RUN_MAX ? first + RUN_MAX : n_full_;
                    while (i < stop && below_r(blob_ + 32 * i)) ++i;
                    if (i < stop) { reduce(blob_ + 32 * i); pending_reduced_ = true; run_ = i + 1; }
                    else if (i < n_full_) run_ = i;
                    else { run_ = n_full_; stage_ = 2; }
                    if (i > first) { s = ShaSeg{blob_ + 32 * first, 32 * (i - first)}; return true; }
                    continue;
                }
                case 2: {                                                                 // ragged tail: right-padded with zeros (helpers.rs:48-52)
                    stage_ = 3;
                    size_t done = n_full_;
                    if (len_ % 32) {
                        uint8_t chunk[32] = {0};
                        memcpy(chunk, blob_ + 32 * n_full_, len_ % 32);
                        reduce(chunk);
                        ++done;
                        zeros_left_ = (n_padded_ - done) * 32;
                        s = ShaSeg{red_, 32};
                        return true;
                    }
                    zeros_left_ = (n_padded_ - done) * 32;
                    continue;
                }
                case 3: {
                    if (!zeros_left_) { stage_ = 4; continue; }
                    static const uint8_t zeros[4096] = {0};
                    const size_t t = zeros_left_ < sizeof zeros ? zeros_left_ : sizeof zeros;
                    zeros_left_ -= t;
                    s = ShaSeg{zeros, t};
                    return true;
                }
                default: return false;
            }
        }
    }
    // 64-byte blocks of the whole prefix (pairing blobs of similar length keeps both streams busy to the end)
    static size_t blocks(size_t n_padded) { return (32 + 32 * n_padded) / 64 + 1; }

private:
#ifdef KZG_TRANSCRIPT_WHOLE_RUNS                                                             // measurement build: the scan as it was (whole runs, memcmp per chunk)
    static constexpr size_t RUN_MAX = ~(size_t)0;
#else
    static constexpr size_t RUN_MAX = 512;                                                // chunks per run: 16 KiB
#endif
    // chunk < r as 32-byte big-endian numbers.  The top eight bytes decide unless they equal r's (2^-64 for random data): one load, one byte swap, one compare per
    // chunk instead of a 32-byte memcmp call.
    bool below_r(const uint8_t* c) const {
#ifdef KZG_TRANSCRIPT_WHOLE_RUNS
        return memcmp(c, r_be_, 32) < 0;
#endif
        uint64_t top;
        memcpy(&top, c, 8);
        top = __builtin_bswap64(top);
        if (top != FR_MODULUS_WORDS[3]) return top < FR_MODULUS_WORDS[3];
        return memcmp(c, r_be_, 32) < 0;
    }
    void reduce(const uint8_t chunk[32]) {                                               // big-endian chunk mod r -> red_ (value < 2^256 < 6 r)
        uint64_t w[4];
        for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int b = 0; b < 8; ++b) v = (v << 8) | chunk[8 * (3 - i) + b]; w[i] = v; }
        while (fr_geq_r(w)) fr_sub_r(w);
        for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) red_[8 * i + b] = (uint8_t)(w[3 - i] >> (8 * (7 - b)));
    }
    const uint8_t* blob_;
    size_t len_, n_full_, n_padded_;
    int stage_ = 0;
    size_t run_ = 0, zeros_left_ = 0;
    bool pending_reduced_ = false;
    uint8_t hdr_[32], red_[32], r_be_[32];
};

inline void sha256_absorb(Sha256& sh, TranscriptPrefix& gen) {
    ShaSeg s;
    while (gen(s)) sha256_update(sh, s.p, s.len);
}

// both generators to exhaustion: whole blocks of the two streams go through the interleaved kernel while both have them, everything else through
// sha256_update (block completion across segments, the tail of the longer stream)
template <class GenA, class GenB>
inline void sha256_absorb_x2(Sha256& a, GenA& gen_a, Sha256& b, GenB& gen_b) {
    ShaSeg sa{nullptr, 0}, sb{nullptr, 0};
    bool ea = false, eb = false;
#if defined(__x86_64__) && !defined(KZG_NO_SHA_X2)      // (KZG_NO_SHA_X2: measurement build, one stream after the other)
    const bool x2 = sha256_have_shani();
#else
    const bool x2 = false;
#endif
    for (;;) {
        while (!ea && sa.len == 0) ea = !gen_a(sa);
        while (!eb && sb.len == 0) eb = !gen_b(sb);
        if (ea && eb) return;
        if (ea || !x2) { if (!eb) { sha256_update(b, sb.p, sb.len); sb.len = 0; } if (!ea) { sha256_update(a, sa.p, sa.len); sa.len = 0; } continue; }
        if (eb) { sha256_update(a, sa.p, sa.len); sa.len = 0; continue; }
        if (a.buf_len || sa.len < 64) {                                                  // bring stream a to a block boundary (or buffer a short segment)
            const size_t t = a.buf_len ? (64 - a.buf_len < sa.len ? 64 - a.buf_len : sa.len) : sa.len;
            sha256_update(a, sa.p, t); sa.p += t; sa.len -= t;
            continue;
        }
        if (b.buf_len || sb.len < 64) {
            const size_t t = b.buf_len ? (64 - b.buf_len < sb.len ? 64 - b.buf_len : sb.len) : sb.len;
            sha256_update(b, sb.p, t); sb.p += t; sb.len -= t;
            continue;
        }
#if defined(__x86_64__) && !defined(KZG_NO_SHA_X2)
        const size_t nb = (sa.len < sb.len ? sa.len : sb.len) / 64;
        sha256_blocks_shani_x2(a.h, sa.p, b.h, sb.p, nb);
        a.total += nb * 64; b.total += nb * 64;
        sa.p += nb * 64; sa.len -= nb * 64; sb.p += nb * 64; sb.len -= nb * 64;
#endif
    }
}

}  // namespace kzg_host
