// naf.h — width-w non-adjacent-form recoding of a 256-bit scalar (host + device): the digit generator of the NAF mode of the
// table-based MSM (msm_kernels.h section "NAF mode", srs.hip srs_build_bit_tables).  Also compiled with g++ by
// tests/hostcheck (tests/test_field29_host.py::test_naf_recoding).
//
// k = sum d_t 2^(pos_t), every d_t odd with |d_t| < 2^(w-1), positions at least w apart: on average one digit per w + 1 bits.
// The consuming recoder: skip the zeros of k + carry (the trailing ONES of k when a negative digit left a carry), take w bits,
// shift; f(pos, (|d| - 1) / 2, d < 0) is called for every digit, low positions first.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_NAF_HD __host__ __device__ __forceinline__
#else
#define KZG_NAF_HD inline
#endif

namespace kzg {

template <class F>
KZG_NAF_HD void naf_for_digits(uint32_t k[8], int w, F&& f) {
    // ONE loop with one exit and one conditional block (the callback), everything else selects.  The first form of this function
    // (a `continue` for the 32-bit skip beside the digit branch, which hipcc turned into nested loops) lost one digit per pair of
    // lanes in k_sort2_scalars when neighbouring lanes took different paths (2^253 - 1 beside 2^36 - 1: found by
    // tests/test_gpu_parity.py::test_msm_adversarial_digit_patterns, located with the host-side sort checker of rounds 3-5).
    const uint32_t mask = (1u << w) - 1u, half = 1u << (w - 1);
    uint32_t pos = 0, carry = 0;
    for (;;) {
        const uint32_t low = carry ? ~k[0] : k[0];
        const bool skip = low == 0;                         // 32 zeros of k + carry: move on by one word
        if (skip && !carry && (k[1] | k[2] | k[3] | k[4] | k[5] | k[6] | k[7]) == 0) break;
        const uint32_t tz = skip ? 0u : (uint32_t)__builtin_ctz(low | 0x80000000u);
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = (uint32_t)((((uint64_t)k[j + 1] << 32) | k[j]) >> tz);
        k[7] >>= tz;
        const uint32_t u = (k[0] & mask) + carry;           // digit: odd; no overflow: carry = 1 only when bit 0 of k is clear
        const uint32_t neg = u > half;
        const uint32_t mag = neg ? (1u << w) - u : u;
        if (!skip) f(pos + tz, (mag - 1u) >> 1, neg);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t shifted = (uint32_t)((((uint64_t)k[j + 1] << 32) | k[j]) >> w);
            k[j] = skip ? k[j + 1] : shifted;
        }
        k[7] = skip ? 0u : k[7] >> w;
        pos += skip ? 32u : tz + (uint32_t)w;
        carry = skip ? carry : neg;
    }
}
#if defined(__HIPCC__)
// Device form used by the sort (k_sort2_scalars<false, true>): the same digits, but the scalar is NOT shifted -- the register form
// above spends two 8-word funnel shifts and eight selects per digit (100 instructions per iteration in the ISA, 2 100 per scalar with
// the conversion to canonical words: 82 us per 2^20 scalars, more than the additions the NAF saves).  Here the thread's 8 words sit in
// LDS (word j at col[j * stride], words 8 .. 10 zero, written by the caller; columns are thread-private, bank = lane: no
// conflicts whatever the word index) and a digit reads the 64-bit window at its position: three ds_read_b32, two funnel shifts.
// Same loop shape as above: one exit, one conditional block.
template <class F>
__device__ __forceinline__ void naf_for_digits_lds(const uint32_t* col, uint32_t stride, int w, F&& f) {
    const uint32_t mask = (1u << w) - 1u, half = 1u << (w - 1);
    uint32_t pos = 0, carry = 0;
    while (pos < 256u || carry) {
        const uint32_t q = pos >> 5, sh = pos & 31u;
        const uint32_t* p = col + (q < 8u ? q : 8u) * stride;
        const uint32_t w0 = p[0], w1 = p[stride], w2 = p[2 * stride];
        const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh);
        const uint32_t hi = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
        const uint32_t low = carry ? ~lo : lo;
        const bool skip = low == 0;                          // 32 zeros of k + carry
        const uint32_t tz = skip ? 0u : (uint32_t)__builtin_ctz(low | 0x80000000u);
        const uint32_t x = (uint32_t)((((uint64_t)hi << 32) | lo) >> tz);
        const uint32_t u = (x & mask) + carry;
        const uint32_t neg = u > half;
        const uint32_t mag = neg ? (1u << w) - u : u;
        if (!skip) f(pos + tz, (mag - 1u) >> 1, neg);
        pos += skip ? 32u : tz + (uint32_t)w;
        carry = skip ? carry : neg;
    }
}
#endif

// Bucket of a NAF digit with key = (|d| - 1) / 2 < 2^nbits: the key rotated right by six bits.  The leftover top bits of a scalar end
// in a SHORT last digit, so small keys are heavy (key 0 collects ~11 % of the scalars' last digits, key k about 1 / k of that); in
// natural order they would all sit in the first group of 64 buckets -- one workgroup of the first reduction level and one coarse
// bin of the sort.  Rotated, the keys 0 .. 63 land 2^(nbits - 6) buckets apart: one per group.  The reduction is indifferent: it
// returns one sum per bit of the bucket index, and the host epilogue gives bit t the weight of the key bit it came from.
KZG_NAF_HD uint32_t naf_bucket(uint32_t key, int nbits) { return ((key & 63u) << (nbits - 6)) | (key >> 6); }
// key bit held by bit t of the bucket index
KZG_NAF_HD int naf_key_bit_of_bucket_bit(int t, int nbits) { return t < nbits - 6 ? t + 6 : t - (nbits - 6); }
constexpr int NAF_DIGITS = 16;                 // words per scalar in the digit array: covers w >= 16 (254 / 16 + 1)
constexpr uint32_t NAF_NO_DIGIT = 0xFFFFFFFFu; // behind a scalar's last digit (a digit word has bits 24..30 clear)
constexpr uint32_t NAF_POSITIONS = 255;        // digit positions 0 .. 254 of a scalar < 2^254
// most entries one scalar can produce in width-w NAF (digits are >= w positions apart)
KZG_NAF_HD int naf_max_digits(int w) { return 254 / w + 1; }

}  // namespace kzg
