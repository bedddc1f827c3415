// poly.hip — the O(n) polynomial work of KZG::compute_proof_impl (prover/src/kzg.rs:128-178) on the GPU:
//   y   = p(z)                    barycentric, primitives/src/helpers.rs:475-535 (incl. the z-on-domain early return :497-504)
//   q_i = (f_i - y) / (w^i - z)   kzg.rs:151-174
//   q_m = sum_{i != m} (f_i - y) w^i / (z (z - w^i))   when z = w^m, kzg.rs:237-260
// then the quotient is committed as MSM(srs, IFFT(q)) (see kzg_commit_eval_form).
// The reference performs one field inversion per division (2n-3n serial inversions); here all denominators w^i - z
// share ONE inversion, done on the host: the domain is a group, so the product of the denominators over a coset
// {j + k n/m : k < m} is known in closed form,
//     prod_k (w^(j + k n/m) - z) = (-1)^(m+1) (W^j - z^m),   W = w^m  (the m-th roots of W^j are exactly those w^..),
// i.e. it is again a denominator of the SAME problem on the m-times smaller domain at the point z^m.  Montgomery's trick needs
// the inverse of each group's product: that is the smaller problem's answer.  The recursion ends at 1 / (1 - z^n), which the
// host computes while it enqueues (one 254-bit exponentiation, ~15 us); the device only multiplies: ~3 products per element
// and level.  (Round 1 / first half of round 2: one Fermat inversion per lane, 381 dependent multiplies = 213 us on the
// critical path of every proof, whatever n.)  Results are identical field elements.
// Also: helpers::calculate_roots_of_unity (helpers.rs:553-589) as a kernel.
#include "poly_common.h"
#include "fe_invert.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace kzg {


struct ProofScalars {        // device-resident small state of one proof computation
    uint32_t on_domain_index; // NO_INDEX if z is not a domain element
    uint32_t pad[3];
    int32_t y[NL];            // y = p(z), internal form, reduced
    uint32_t y_wire[8];
};

// ---- K0: the inverses 1 / (W^j - Z) on a small domain, one workgroup ----------------------------------------------
// out[j] = 1 / (w^(j e) - z^e), j < N = 2^log_ns <= 4096, e = n / N; zt[a] = z^(2^a) (wire), zt[log_n + 1] = 1 / (1 - z^n).
// Level by level from the single value 1 / (1 - z^n), size s -> 2 s, in place in LDS: lane t < s owns the pair {t, t + s} of the
// 2s-point domain, whose elements are +-W^t:  d0 = W^t - Z, d1 = -W^t - Z, d0 d1 = -(W^2t - Z^2), so with g = the value of the
// coarser level  1/d0 = -g d1,  1/d1 = -g d0: two independent multiplies per level and lane, log2 N levels.
// A zero denominator (z on the domain, the case of compute_proof_with_known_z_fr_index) gets the inverse 1, as the callers expect;
// the other denominators of its group are Z (rho - 1) for the roots of unity rho != 1 of the group, so their inverses are Z^-1 times
// constants: the host supplies z^-(2^a) and 1/(i-1), -1/2, 1/(-i-1) with the other scalars (zt[log_n+2 ..]).  The value such a lane
// read from the coarser level was never valid and is not used.
constexpr int POLY_SMALL_THREADS = 1024;
constexpr uint32_t POLY_SMALL_MAX_LOG = 12;
constexpr uint32_t POLY_SMALL_MAX = 1u << POLY_SMALL_MAX_LOG;
__global__ void __launch_bounds__(POLY_SMALL_THREADS)
k_poly_inv_small(const uint4* __restrict__ zt, int log_n, int log_ns, NttTables tb, int32_t* __restrict__ out /* planes, stride 2^log_ns */) {
    extern __shared__ int32_t lds_inv[];                      // NL planes of 2^log_ns
    __shared__ int32_t zl[(POLY_SMALL_MAX_LOG + 1) * NL];     // canonical z^(2^log_e) of every level
    const uint32_t t0 = threadIdx.x, N = 1u << log_ns;
    if (t0 == 0) {
        Fr top;
        wire_load(top, zt, (size_t)log_n + 1);
#pragma unroll
        for (int j = 0; j < NL; ++j) lds_inv[(size_t)j * N] = top.l[j];
    }
    if ((int)t0 < log_ns) {                                   // level log_s = t0 works on the 2^(t0+1)-point domain
        Fr z;
        wire_load(z, zt, (size_t)(log_n - (int)t0 - 1));
        fe_canon(z);
#pragma unroll
        for (int j = 0; j < NL; ++j) zl[t0 * NL + j] = z.l[j];
    }
    __syncthreads();
    for (int log_s = 0; log_s < log_ns; ++log_s) {
        const uint32_t s = 1u << log_s;
        const int log_e = log_n - log_s - 1;                   // the 2s-point domain: generator w^(2^log_e), point z^(2^log_e)
        // part A, independent of the coarser level: the two denominators of each of this thread's lanes (at most two lanes)
        Fr d0[2], d1[2];
        Fr z;
#pragma unroll
        for (int j = 0; j < NL; ++j) z.l[j] = zl[log_s * NL + j];
        auto part_a = [&](int q) {
            const uint32_t t = t0 + q * POLY_SMALL_THREADS;
            if (t >= s) return;
            Fr w;
            domain_elem(w, tb, t << log_e);
            fe_canon(w);
            fe_sub(d0[q], w, z);                               // W^t - Z
            fe_neg(d1[q], w);
            fe_norm(d1[q]);
            fe_canon(d1[q]);                                   // -W^t, canonical
            fe_sub(d1[q], d1[q], z);
        };
        part_a(0);
        part_a(1);
        // part B
        auto part_b = [&](int q) {
            const uint32_t t = t0 + q * POLY_SMALL_THREADS;
            if (t >= s) return;
            Fr g, i0, i1;
#pragma unroll
            for (int j = 0; j < NL; ++j) g.l[j] = -lds_inv[(size_t)j * N + t];
            fe_norm(g);
            const bool z0 = fe_is_literal_zero(d0[q]), z1 = fe_is_literal_zero(d1[q]);
            if (__builtin_expect(z0 || z1, 0)) {               // +-W^t = Z: the other denominator is -2 Z, its inverse (-1/2) Z^-1 from the host's table
                Fr zi, c2, other;
                wire_load(zi, zt, (size_t)(log_n + 2 + log_e));
                wire_load(c2, zt, (size_t)(2 * log_n + 3 + 1));
                fe_mul(other, zi, c2);
                fe_set_one(i0); fe_set_one(i1);
                if (!z0) i0 = other;
                if (!z1) i1 = other;
            } else {
                fe_mul2(i0, g, d1[q], i1, g, d0[q]);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) { lds_inv[(size_t)j * N + t] = i0.l[j]; lds_inv[(size_t)j * N + t + s] = i1.l[j]; }
        };
        part_b(0);
        part_b(1);
        __syncthreads();
    }
    for (uint32_t i = t0; i < N; i += POLY_SMALL_THREADS)
#pragma unroll
        for (int j = 0; j < NL; ++j) out[(size_t)j * N + i] = lds_inv[(size_t)j * N + i];
}

// The four inverses of the coset {t + k T : k < 4} of the 4T-point domain (generator w^(2^log_e), point Z = z^(2^log_e), canonical):
// its elements are W^t times the fourth roots of unity 1, i, -1, -i (i = w^(n/4)), and d0 d1 d2 d3 = -(W^4t - Z^4), so with
// G = -next[t]:  1/d0 = G d1 (d2 d3),  1/d1 = G d0 (d2 d3),  1/d2 = G (d0 d1) d3,  1/d3 = G (d0 d1) d2.
// Returns the index k of a zero denominator (z on the domain; its inverse is set to 1), or 4.
__device__ __forceinline__ uint32_t inv4_group(Fr inv[4], const NttTables& tb, const uint4* __restrict__ zt, int log_n, int log_e, uint32_t t, const Fr& z,
                                               const int32_t* __restrict__ next, size_t next_stride) {
    Fr w0, w1, qi, d[4];
    domain_elem(w0, tb, t << log_e);
    domain_elem(qi, tb, 1u << (log_n - 2));                    // w^(n/4): the primitive fourth root
    fe_mul(w1, w0, qi);
    fe_canon(w0);
    fe_canon(w1);
    fe_sub(d[0], w0, z);
    fe_sub(d[1], w1, z);
    Fr n0, n1;
    fe_neg(n0, w0); fe_norm(n0); fe_canon(n0);
    fe_neg(n1, w1); fe_norm(n1); fe_canon(n1);
    fe_sub(d[2], n0, z);
    fe_sub(d[3], n1, z);
    uint32_t zero_k = 4;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) if (fe_is_literal_zero(d[k])) { zero_k = k; fe_set_one(d[k]); }
    if (__builtin_expect(zero_k != 4, 0)) {                    // d_k = Z (i^(k - k0) - 1): inverses = Z^-1 times the host's constants
        Fr zi;
        wire_load(zi, zt, (size_t)(log_n + 2 + log_e));
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t r = (k - zero_k) & 3u;
            if (r == 0) { fe_set_one(inv[k]); continue; }
            Fr c;
            wire_load(c, zt, (size_t)(2 * log_n + 3 + (r - 1)));
            fe_mul(inv[k], zi, c);
        }
        return zero_k;
    }
    Fr p01, p23, G;
    fe_mul2(p01, d[0], d[1], p23, d[2], d[3]);
    pl_load(G, next, next_stride, t);
    fe_neg(G, G);
    fe_norm(G);
    Fr a, b;
    fe_mul2(a, G, p23, b, G, p01);
    fe_mul2(inv[0], a, d[1], inv[1], a, d[0]);
    fe_mul2(inv[2], b, d[3], inv[3], b, d[2]);
    return zero_k;
}

// ---- K0b: one x4 level for domains too large for K0 --------------------------------------------------------------------
// out[j] = 1 / (w^(j e) - z^e), j < N = 2^log_nl, e = n / N, from next[t] = 1 / (w^(4 t e) - z^(4 e)), t < N / 4
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_inv_level(const uint4* __restrict__ zt, int log_n, int log_nl, NttTables tb, const int32_t* __restrict__ next, int32_t* __restrict__ out) {
    const uint32_t N = 1u << log_nl, T = N >> 2, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int log_e = log_n - log_nl;
    Fr z, inv[4];
    wire_load(z, zt, (size_t)log_e);
    fe_canon(z);
    inv4_group(inv, tb, zt, log_n, log_e, t, z, next, T);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) pl_store(out, N, t + k * T, inv[k]);
}

// y = (z^n - 1) / n * sum  for z off the domain (helpers.rs:529-532), into the proof's scalar image; one thread
__device__ __forceinline__ void poly_store_y_off_domain(const Fr& sum, int log_n, const uint4* __restrict__ z_wire, ProofScalars* __restrict__ ps) {
    Fr y, z, zn, one, ninv;
    wire_load(z, z_wire, 0);
    zn = z;
    for (int k = 0; k < log_n; ++k) fe_sqr(zn, zn);          // z^n, n = 2^log_n
    fe_set_one(one);
    fe_sub(zn, zn, one);                           // in (-3m, 2m)
#pragma unroll
    for (int j = 0; j < NL; ++j) ninv.l[j] = (int32_t)FrParams::NINV[log_n * NL + j];
    fe_mul(y, sum, zn);
    fe_mul(y, y, ninv);
#pragma unroll
    for (int j = 0; j < NL; ++j) ps->y[j] = y.l[j];
    fe_to_wire(ps->y_wire, y);
}

// ---- K1: the last level (inverses of all n denominators) + barycentric partial sums ---------------------------------------
// lane t owns elements i = k * T + t.  inv[i] = 1 / (w^i - z)  (1 for the on-domain index).
//   direct != 0: next[i] already is that inverse (n <= 4096: K0 produced all of them); any number of lanes
//   direct == 0: T = n / 4 lanes and next[t] = 1 / (w^(4 t) - z^4) (inv4_group)
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_inverses(const uint4* __restrict__ evals, uint32_t n, int log_n, NttTables tb, const uint4* __restrict__ z_wire,
                const int32_t* __restrict__ next, int direct, int32_t* __restrict__ inv, int32_t* __restrict__ partial /* NL x gridDim */,
                ProofScalars* __restrict__ ps, int finish_y /* one workgroup, z off the domain: it also does what k_poly_finish_y would */) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr z;
    wire_load(z, z_wire, 0);
    fe_canon(z);
    Fr sum;
    fe_set_zero(sum);
    if (direct) {
        uint32_t cnt = 0;
        for (uint32_t i = t; i < n; i += T, ++cnt) {
            Fr w, wc, d, iv, f, term;
            domain_elem(w, tb, i);
            wc = w;
            fe_canon(wc);
            fe_sub(d, wc, z);
            if (fe_is_literal_zero(d)) ps->on_domain_index = i;
            pl_load(iv, next, n, i);
            if (next != inv) pl_store(inv, n, i, iv);
            wire_load(f, evals, i);
            fe_mul(term, f, w);
            fe_mul(term, term, iv);
            fe_sub(sum, sum, term);                   // barycentric term f_i w^i / (z - w^i) = -(f_i w^i inv_i)
            fe_norm(sum);                             // |sum| grows by 2m per term
            if ((cnt & 31u) == 31u) fe_reduce(sum);
        }
    } else if (t < (n >> 2)) {
        const uint32_t Tq = n >> 2;
        Fr iv[4], f[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) wire_load(f[k], evals, t + k * Tq);          // in flight during the inversion arithmetic
        const uint32_t zero_k = inv4_group(iv, tb, z_wire, log_n, 0, t, z, next, Tq);
        if (zero_k != 4) ps->on_domain_index = t + zero_k * Tq;
        Fr w0, w1, qi;
        domain_elem(w0, tb, t);
        domain_elem(qi, tb, 1u << (log_n - 2));
        fe_mul(w1, w0, qi);
        Fr fw[4];
        fe_mul2(fw[0], f[0], w0, fw[2], f[2], w0);                                     // w^(t + 2 Tq) = -w^t
        fe_mul2(fw[1], f[1], w1, fw[3], f[3], w1);
        Fr tm[4];
        fe_mul2(tm[0], fw[0], iv[0], tm[2], fw[2], iv[2]);
        fe_mul2(tm[1], fw[1], iv[1], tm[3], fw[3], iv[3]);
        // sum = -(tm0 + tm1) + (tm2 + tm3): elements 2 and 3 carry the factor -1 of their root
        fe_sub(sum, tm[2], tm[0]);
        fe_add(sum, sum, tm[3]);
        fe_sub(sum, sum, tm[1]);
        fe_norm(sum);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) pl_store(inv, n, t + k * Tq, iv[k]);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x != 0) return;
    if (finish_y) poly_store_y_off_domain(sum, log_n, z_wire, ps);
    else pl_store(partial, gridDim.x, blockIdx.x, sum);
}

// ---- K2: y ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_finish_y(const uint4* __restrict__ evals, uint32_t n, int log_n, const uint4* __restrict__ z_wire,
                const int32_t* __restrict__ partial, uint32_t n_partial, ProofScalars* __restrict__ ps) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    Fr sum;
    fe_set_zero(sum);
    for (uint32_t i = threadIdx.x; i < n_partial; i += POLY_THREADS) {
        Fr v;
        pl_load(v, partial, n_partial, i);
        fe_add(sum, sum, v);
        fe_norm(sum);
        if ((i / POLY_THREADS) % 32 == 31) fe_reduce(sum);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x != 0) return;
    Fr y;
    uint32_t m = ps->on_domain_index;
    if (m != NO_INDEX) {
        wire_load(y, evals, m);                   // helpers.rs:497-504
    } else {
        poly_store_y_off_domain(sum, log_n, z_wire, ps);
        return;
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) ps->y[j] = y.l[j];
    fe_to_wire(ps->y_wire, y);
}

// ---- K3: quotient evaluations -----------------------------------------------------------------------------
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_quotient(const uint4* __restrict__ evals, uint32_t n, NttTables tb, const int32_t* __restrict__ inv,
                const ProofScalars* __restrict__ ps, uint4* __restrict__ q_out, int32_t* __restrict__ partial) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = ps->on_domain_index;
    Fr y;
#pragma unroll
    for (int j = 0; j < NL; ++j) y.l[j] = ps->y[j];
    Fr sum;
    fe_set_zero(sum);
    uint32_t cnt = 0;
    for (uint32_t i = t; i < n; i += T, ++cnt) {
        if (i == m) continue;
        Fr f, iv, q;
        wire_load(f, evals, i);
        pl_load(iv, inv, n, i);
        fe_sub(f, f, y);                          // (-3m, 3m)
        fe_mul(q, f, iv);
        wire_store(q_out, i, q);
        if (m != NO_INDEX) {                      // kzg.rs:237-260: accumulate q_i w^i
            Fr w, term;
            domain_elem(w, tb, i);
            fe_mul(term, q, w);
            fe_add(sum, sum, term);
            fe_norm(sum);
            if (cnt % 32 == 31) fe_reduce(sum);
        }
    }
    if (m == NO_INDEX) return;                    // uniform across the grid
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x == 0) pl_store(partial, gridDim.x, blockIdx.x, sum);
}

// ---- K4: on-domain element q_m = -(1/z) sum_{i != m} q_i w^i, 1/z = w^(n-m) -------------------------------------
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_quotient_on_domain(uint32_t n, NttTables tb, const int32_t* __restrict__ partial, uint32_t n_partial,
                          const ProofScalars* __restrict__ ps, uint4* __restrict__ q_out) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t m = ps->on_domain_index;
    if (m == NO_INDEX) return;
    Fr sum;
    fe_set_zero(sum);
    for (uint32_t i = threadIdx.x; i < n_partial; i += POLY_THREADS) {
        Fr v;
        pl_load(v, partial, n_partial, i);
        fe_add(sum, sum, v);
        fe_norm(sum);
        if ((i / POLY_THREADS) % 32 == 31) fe_reduce(sum);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x != 0) return;
    Fr zinv, q;
    domain_elem(zinv, tb, (n - m) & (n - 1));
    fe_mul(q, sum, zinv);
    fe_neg(q, q);
    fe_norm(q);
    wire_store(q_out, m, q);
}

// ---- K3' / K4': z = w^m with m known on the host (compute_proof_with_known_z_fr_index, kzg.rs:237-260), n <= POLY_SMALL_MAX ---------
// The denominators of an on-domain point are the SAME set for every m:  w^i - w^m = w^m (w^(i-m) - 1), so with the per-domain table
// t1[k] = 1 / (w^k - 1) (built once per domain size by the inversion chain above at z = 1) the inverse is w^(n-m) t1[(i - m) mod n]: no
// inversion chain, no barycentric sum (y = f_m, helpers.rs:497-504): three kernels (46 + 15 + 6 us for 2 048 evaluations) less per proof.
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_quotient_table(const uint4* __restrict__ evals, uint32_t n, NttTables tb, const int32_t* __restrict__ t1, uint32_t m,
                      uint4* __restrict__ q_out, int32_t* __restrict__ partial, ProofScalars* __restrict__ ps /* one workgroup: it also finishes q_m and y */) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr y, zinv;
    wire_load(y, evals, m);
    fe_reduce(y);
    domain_elem(zinv, tb, (n - m) & (n - 1));             // w^-m
    Fr sum;
    fe_set_zero(sum);
    uint32_t cnt = 0;
    for (uint32_t i = t; i < n; i += T, ++cnt) {
        if (i == m) continue;
        Fr f, iv, q, w, term;
        wire_load(f, evals, i);
        pl_load(iv, t1, n, (i - m) & (n - 1));
        fe_mul(iv, iv, zinv);                             // 1 / (w^i - w^m)
        fe_sub(f, f, y);                                  // (-3m, 3m)
        fe_mul(q, f, iv);
        wire_store(q_out, i, q);
        domain_elem(w, tb, i);
        fe_mul(term, q, w);
        fe_add(sum, sum, term);
        fe_norm(sum);
        if (cnt % 32 == 31) fe_reduce(sum);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x != 0) return;
    if (gridDim.x > 1) { pl_store(partial, gridDim.x, blockIdx.x, sum); return; }
    Fr q;                                                 // n <= 1 024: the only workgroup -- what k_poly_quotient_on_domain_known would do
    fe_mul(q, sum, zinv);
    fe_neg(q, q);
    fe_norm(q);
    wire_store(q_out, m, q);
    ps->on_domain_index = m;
#pragma unroll
    for (int j = 0; j < NL; ++j) ps->y[j] = y.l[j];
    fe_to_wire(ps->y_wire, y);
}
// as K4, and y = f_m for the read-back
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_quotient_on_domain_known(const uint4* __restrict__ evals, uint32_t n, NttTables tb, const int32_t* __restrict__ partial, uint32_t n_partial,
                                uint32_t m, ProofScalars* __restrict__ ps, uint4* __restrict__ q_out) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    Fr sum;
    fe_set_zero(sum);
    for (uint32_t i = threadIdx.x; i < n_partial; i += POLY_THREADS) {
        Fr v;
        pl_load(v, partial, n_partial, i);
        fe_add(sum, sum, v);
        fe_norm(sum);
        if ((i / POLY_THREADS) % 32 == 31) fe_reduce(sum);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x != 0) return;
    Fr zinv, q, y;
    domain_elem(zinv, tb, (n - m) & (n - 1));
    fe_mul(q, sum, zinv);
    fe_neg(q, q);
    fe_norm(q);
    wire_store(q_out, m, q);
    wire_load(y, evals, m);
    ps->on_domain_index = m;
#pragma unroll
    for (int j = 0; j < NL; ++j) ps->y[j] = y.l[j];
    fe_to_wire(ps->y_wire, y);
}

// ---- blob bytes -> Fr (helpers::to_fr_array, primitives/src/helpers.rs:40-57) ---------------------------------------
// element i = the 32 big-endian bytes [32 i, 32 i + 32) (the last chunk right-padded with zeros) mod r, emitted in wire
// (Montgomery, radix 2^256) form; elements i >= n_elems (power-of-two padding of PolynomialEvalForm::new,
// polynomial.rs:49-51) are zero.  One multiply: x * (2^(256+261) mod r) * 2^-261 = x * 2^256 mod r, x < 2^256 < 5.3 r.
__global__ void __launch_bounds__(POLY_THREADS)
k_blob_to_fr(const uint8_t* __restrict__ bytes, size_t len, uint32_t n_elems, uint32_t n_padded, uint4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    uint32_t w32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n_elems && (size_t)i * 32 + 32 <= len) {
        // a whole chunk (all but a ragged last one): two 16-byte loads and byte swaps instead of 32 single-byte loads -- 67 -> ~20 us at 2^20 elements
        // (the staging buffer comes from hipMalloc: 16-byte aligned)
        const uint4* p = reinterpret_cast<const uint4*>(bytes + (size_t)i * 32);
        const uint4 hi = p[0], lo = p[1];          // bytes 0..15 (most significant), 16..31
        w32[7] = __builtin_bswap32(hi.x); w32[6] = __builtin_bswap32(hi.y); w32[5] = __builtin_bswap32(hi.z); w32[4] = __builtin_bswap32(hi.w);
        w32[3] = __builtin_bswap32(lo.x); w32[2] = __builtin_bswap32(lo.y); w32[1] = __builtin_bswap32(lo.z); w32[0] = __builtin_bswap32(lo.w);
    } else if (i < n_elems) {
        const size_t base = (size_t)i * 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) {              // word k (little-endian) = bytes 28-4k .. 31-4k of the big-endian chunk
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                size_t pos = base + (size_t)(28 - 4 * k + b);
                uint32_t byte = pos < len ? bytes[pos] : 0u;
                w = (w << 8) | byte;
            }
            w32[k] = w;
        }
    }
    Fr x, kk, r;
    fe_unpack(x, w32);
#pragma unroll
    for (int j = 0; j < NL; ++j) kk.l[j] = (int32_t)FrParams::K_RAW[j];       // 2^(256+261) mod r
    fe_mul(r, x, kk);                                                          // x * 2^256 mod r, in (-m, 2m)
    fe_canon(r);
    uint32_t o[8];
    fe_pack(o, r);
    out[2 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[2 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// ---- roots of unity -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_roots(uint4* __restrict__ out, uint32_t n, NttTables tb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr w;
    domain_elem(w, tb, i);
    wire_store(out, i, w);
}

// ---- batch verification front end: all n barycentric evaluations of verify_blob_kzg_proof_batch in two launches ---------------
// verifier/src/batch.rs:16-69 -> primitives/src/helpers.rs:613-662 evaluates y_i = p_i(z_i) blob after blob (n_i inversions each).
// Here blob i (2^log_n <= 2^VB_MAX_LOG evaluations, read straight from its big-endian bytes) is one workgroup:
//   k_vb_prep   one LANE per blob: z^n by squarings and the ONE inversion of the blob, 1 / (z^n - 1) = 1 / prod_j (z - w^j) (Fermat)
//   k_vb_eval   one WORKGROUP per blob: lane t owns the denominators d_j = z - w^j, j = t + k Lf; a product tree over the lanes
//               in LDS (up-sweep), seeded at the root with the inverse from k_vb_prep and walked down (inverse of a node =
//               inverse of its parent x product of its sibling) leaves every lane the inverse of its own product; Montgomery's
//               trick inside the lane; then sum_j f_j w^j / (z - w^j) and y = (z^n - 1) / n times that sum (helpers.rs:507-532).
// z on the domain (1 - z^n == 0; the early return of helpers.rs:497-504) and blobs beyond 2^VB_MAX_LOG elements are flagged and evaluated by
// the single-polynomial path (proof_run) on the host's request.
constexpr int VB_MAX_LOG = 12;
constexpr int VB_THREADS = 512;         // 8 waves: two per SIMD, 256 VGPRs each (the asm products keep inputs and outputs apart)
struct VbBlob { uint64_t off; uint32_t len; uint32_t log_n; };          // byte offset (32-byte aligned, zero-filled to the next chunk), byte length, log2(padded elements)
struct VbPrep { int32_t z[NL]; int32_t znm1[NL]; int32_t tinv[NL]; uint32_t fallback; };

__global__ void __launch_bounds__(64)
k_vb_prep(const uint4* __restrict__ zs_wire, const VbBlob* __restrict__ meta, uint32_t nb, VbPrep* __restrict__ prep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int log_n = (int)meta[i].log_n;
    Fr z, zn, one, den;
    wire_load(z, zs_wire, i);
    zn = z;
    for (int a = 0; a < log_n; ++a) fe_sqr(zn, zn);
    fe_set_one(one);
    fe_sub(den, zn, one);
    fe_reduce(den);                                        // z^n - 1 in (-m, 2m)
    Fr dc = den;
    fe_canon(dc);
    VbPrep& o = prep[i];
    o.fallback = (fe_is_literal_zero(dc) || log_n > VB_MAX_LOG) ? 1u : 0u;
    fe_canon(z);
#pragma unroll
    for (int j = 0; j < NL; ++j) { o.z[j] = z.l[j]; o.znm1[j] = den.l[j]; }
    if (o.fallback) return;
    // 1 / den by division steps (fe_invert.h, round 4; a^(r-2) before: 380 dependent products on a lone lane)
    Fr acc;
#if !defined(KZG_INVERT_FERMAT)
    fe_inverse_safegcd(acc, den);
#else
    acc = den;
#pragma unroll 1
    for (int bit = 252; bit >= 0; --bit) {
        fe_sqr(acc, acc);
        const uint32_t limb = FrParams::P[bit / LB] - (bit / LB == 0 ? 2u : 0u);
        if ((limb >> (bit % LB)) & 1u) fe_mul(acc, acc, den);
    }
#endif
#pragma unroll
    for (int j = 0; j < NL; ++j) o.tinv[j] = acc.l[j];
}

// raw big-endian chunk j of a packed blob as a plain integer < 2^256 in limbs (zero beyond the blob's elements)
__device__ __forceinline__ void vb_load_raw(Fr& x, const uint8_t* __restrict__ base, uint32_t j, uint32_t n_elems) {
    uint32_t w32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j < n_elems) {
        const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)j * 32);
        const uint4 hi = p[0], lo = p[1];                  // bytes 0..15 (most significant), 16..31
        w32[7] = __builtin_bswap32(hi.x); w32[6] = __builtin_bswap32(hi.y); w32[5] = __builtin_bswap32(hi.z); w32[4] = __builtin_bswap32(hi.w);
        w32[3] = __builtin_bswap32(lo.x); w32[2] = __builtin_bswap32(lo.y); w32[1] = __builtin_bswap32(lo.z); w32[0] = __builtin_bswap32(lo.w);
    }
    fe_unpack(x, w32);
}

// lane t of a blob's workgroup: its M denominators, their product into the tree leaf (phase 1), and after the tree walk the M
// barycentric terms (phase 2).  M = n / Lf in {1, 2, 4, 8}: a template parameter so that d[] / pre[] stay in registers.
template <int M>
struct VbLane {
    Fr d[M], pre[M];
    __device__ __forceinline__ void denominators(Fr& p, const Fr& z, const NttTables& tb, uint32_t t, uint32_t Lf, int sh) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
            Fr w;
            domain_elem(w, tb, (t + (uint32_t)k * Lf) << sh);
            fe_canon(w);
            fe_sub(d[k], z, w);                            // canonical - canonical: limbs within +-2^29, |d| < m
            if (k == 0) p = d[0];
            else { pre[k] = p; fe_mul(p, p, d[k]); }
        }
    }
    __device__ __forceinline__ void terms(Fr& sum, Fr inv_all, const NttTables& tb, const uint8_t* __restrict__ base, uint32_t n_elems, uint32_t t, uint32_t Lf, int sh) {
#pragma unroll
        for (int k = M - 1; k >= 0; --k) {
            Fr inv;
            if (k > 0) { fe_mul(inv, inv_all, pre[k]); fe_mul(inv_all, inv_all, d[k]); }
            else inv = inv_all;
            const uint32_t jdx = t + (uint32_t)k * Lf;
            Fr x, w, term;
            vb_load_raw(x, base, jdx, n_elems);
            domain_elem(w, tb, jdx << sh);
            fe_mul(term, x, w);                            // plain integer x internal form = plain residue f_j w^j
            fe_mul(term, term, inv);
            fe_add(sum, sum, term);
            if (M > 4 && (k & 1) == 0) fe_norm(sum);       // at most four unnormalised 29-bit limbs fit an int32: keep the running sum normalised
        }
        fe_norm(sum);                                      // <= 8 terms of (-m, 2m)
        fe_reduce(sum);
    }
};

template <int M>
__device__ __forceinline__ void vb_eval_body(int32_t* __restrict__ tree, const uint8_t* __restrict__ bytes, const VbBlob& b, const VbPrep& pr,
                                             const NttTables& tb, uint4* __restrict__ ys_wire) {
    const int log_n = (int)b.log_n;
    const uint32_t n = 1u << log_n, Lf = n / (uint32_t)M, S = 2 * Lf;
    const uint32_t t = threadIdx.x, n_elems = (b.len + 31) / 32;
    const int sh = VB_MAX_LOG - log_n;                     // w_n^j = w_4096^(j << sh)
    Fr z;
#pragma unroll
    for (int j = 0; j < NL; ++j) z.l[j] = pr.z[j];
    VbLane<M> lane;
    if (t < Lf) {
        Fr p;
        lane.denominators(p, z, tb, t, Lf, sh);
#pragma unroll
        for (int j = 0; j < NL; ++j) tree[j * S + Lf + t] = p.l[j];
    }
    __syncthreads();
    for (uint32_t s = Lf >> 1; s >= 1; s >>= 1) {          // up-sweep: node = product of its two children
        if (t < s) {
            const uint32_t node = s + t;
            Fr a, c, r;
#pragma unroll
            for (int j = 0; j < NL; ++j) { a.l[j] = tree[j * S + 2 * node]; c.l[j] = tree[j * S + 2 * node + 1]; }
            fe_mul(r, a, c);
#pragma unroll
            for (int j = 0; j < NL; ++j) tree[j * S + node] = r.l[j];
        }
        __syncthreads();
    }
    if (t == 0) {
#pragma unroll
        for (int j = 0; j < NL; ++j) tree[j * S + 1] = pr.tinv[j];                   // 1 / prod_j (z - w^j) = 1 / (z^n - 1)
    }
    __syncthreads();
    for (uint32_t s = 1; s < Lf; s <<= 1) {                // down-sweep: inverse of a child = inverse of the node x its sibling
        if (t < s) {
            const uint32_t node = s + t;
            Fr g, a, c, ia, ic;
#pragma unroll
            for (int j = 0; j < NL; ++j) { g.l[j] = tree[j * S + node]; a.l[j] = tree[j * S + 2 * node]; c.l[j] = tree[j * S + 2 * node + 1]; }
            fe_mul2(ia, g, c, ic, g, a);
#pragma unroll
            for (int j = 0; j < NL; ++j) { tree[j * S + 2 * node] = ia.l[j]; tree[j * S + 2 * node + 1] = ic.l[j]; }
        }
        __syncthreads();
    }
    Fr sum;
    fe_set_zero(sum);
    if (t < Lf) {
        Fr inv_all;
#pragma unroll
        for (int j = 0; j < NL; ++j) inv_all.l[j] = tree[j * S + Lf + t];
        lane.terms(sum, inv_all, tb, bytes + b.off, n_elems, t, Lf, sh);
    }
    __syncthreads();                                       // the tree is dead: its planes carry the block sum
    int level = 0;
    for (uint32_t dd = VB_THREADS / 2; dd >= 1; dd >>= 1, ++level) {
#pragma unroll
        for (int j = 0; j < NL; ++j) tree[j * VB_THREADS + t] = sum.l[j];
        __syncthreads();
        if (t < dd) {
            Fr u;
#pragma unroll
            for (int j = 0; j < NL; ++j) u.l[j] = tree[j * VB_THREADS + t + dd];
            fe_add(sum, sum, u);
            fe_norm(sum);
            if (level % 4 == 3) fe_reduce(sum);            // 16 terms of (-m, 2m) since the last reduction
        }
        __syncthreads();
    }
    if (t != 0) return;
    fe_reduce(sum);
    Fr y, znm1, ninv, kraw;
#pragma unroll
    for (int j = 0; j < NL; ++j) { znm1.l[j] = pr.znm1[j]; ninv.l[j] = (int32_t)FrParams::NINV[log_n * NL + j]; kraw.l[j] = (int32_t)FrParams::K_RAW[j]; }
    fe_mul(y, sum, znm1);
    fe_mul(y, y, ninv);                                    // helpers.rs:529-532; y is a PLAIN residue here (the raw f_j carried no Montgomery factor)
    fe_mul(y, y, kraw);                                    // -> wire form y 2^256
    fe_canon(y);
    uint32_t o[8];
    fe_pack(o, y);
    ys_wire[2 * (size_t)blockIdx.x] = make_uint4(o[0], o[1], o[2], o[3]);
    ys_wire[2 * (size_t)blockIdx.x + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

__global__ void __launch_bounds__(VB_THREADS)
k_vb_eval(const uint8_t* __restrict__ bytes, const VbBlob* __restrict__ meta, const VbPrep* __restrict__ prep, NttTables tb /* 2^VB_MAX_LOG domain */,
          uint4* __restrict__ ys_wire) {
    extern __shared__ int32_t tree[];                      // NL planes of 2 Lf nodes (heap order: root 1, leaves Lf + t)
    const VbBlob b = meta[blockIdx.x];
    const VbPrep& pr = prep[blockIdx.x];
    if (pr.fallback) return;                               // uniform across the workgroup
    const uint32_t n = 1u << b.log_n;
    if (n <= (uint32_t)VB_THREADS) vb_eval_body<1>(tree, bytes, b, pr, tb, ys_wire);
    else if (n == 2u * VB_THREADS) vb_eval_body<2>(tree, bytes, b, pr, tb, ys_wire);
    else if (n == 4u * VB_THREADS) vb_eval_body<4>(tree, bytes, b, pr, tb, ys_wire);
    else vb_eval_body<8>(tree, bytes, b, pr, tb, ys_wire);
}

// ---- host -----------------------------------------------------------------------------------------------------
static int ilog2_exact(size_t n) { int k = 0; while (((size_t)1 << k) < n) ++k; return k; }

int32_t roots_run(kzg_ctx* ctx, uint64_t* out, size_t n) {
    int log_n = ilog2_exact(n);
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, log_n, false, &tb);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * 32));
    hipLaunchKernelGGL(k_poly_roots, dim3((unsigned)((n + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, ctx->stream,
                       ctx->poly[0].a.as<uint4>(), (uint32_t)n, tb);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->poly[0].a.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

// bytes (host) -> n_padded wire elements (device).  Default buffers: ctx->poly[0].c (bytes), ctx->poly[0].a (elements), ctx->stream.
int32_t blob_to_fr_run(kzg_ctx* ctx, const uint8_t* bytes, size_t len, size_t n_padded, void** d_out,
                       hipStream_t st, DeviceBuffer* d_bytes, DeviceBuffer* d_elems) {
    if (!st) st = ctx->stream;
    if (!d_bytes) d_bytes = &ctx->poly[0].c;
    if (!d_elems) d_elems = &ctx->poly[0].a;
    const size_t n_elems = (len + 31) / 32;
    KZG_HIP_TRY(ctx, d_elems->reserve(n_padded * 32 + 32));
    KZG_HIP_TRY(ctx, d_bytes->reserve(len + 32));
    if (len) KZG_HIP_TRY(ctx, hipMemcpyAsync(d_bytes->p, bytes, len, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_blob_to_fr, dim3((unsigned)((n_padded + POLY_THREADS - 1) / POLY_THREADS)), dim3(POLY_THREADS), 0, st,
                       d_bytes->as<uint8_t>(), len, (uint32_t)n_elems, (uint32_t)n_padded, d_elems->as<uint4>());
    KZG_HIP_TRY(ctx, hipGetLastError());
    *d_out = d_elems->p;
    return KZG_OK;
}


// Enqueue the O(n) part of a proof on `st` with the buffers of `ps_set`, without waiting: upload, denominators + batch
// inversion, y, quotient (+ on-domain entry), IFFT of the quotient.  y is copied back into ps_set.pinned + 2048 (valid once the
// stream has been synchronised); the quotient's coefficients are left in ps_set.c.
// skip_intt: the caller commits the quotient's EVALUATIONS over a Lagrange basis (prover/src/kzg.rs:96-100 applied to the quotient, exactly
// what the reference's compute_proof_impl does: kzg.rs:176-177), so they stay in set.c as they are.
// d_resident: the n evaluations (wire) already on the device in a buffer of the caller's (the blob stream's jobs, capi.hip); read in place
static int32_t proof_enqueue(kzg_ctx* ctx, PolySet& set, hipStream_t st, NttWorkspace* nttws, const uint64_t* evals, size_t n,
                             const uint64_t z[4], bool want_proof, bool skip_intt = false, const uint4* d_resident = nullptr) {
    RoctxRange range(want_proof ? "kzg:proof:inverses + y + quotient + intt" : "kzg:evaluate:inverses + y");
    int log_n = ilog2_exact(n);
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, log_n, false, &tb);
    if (rc != KZG_OK) return rc;
    if (want_proof && n > 1 && !skip_intt) { NttTables tbi; rc = ntt_get_tables(ctx, log_n, true, &tbi); if (rc != KZG_OK) return rc; }
    // lanes of the last level: 4 elements each (one coset), at least one block
    const int per_lane = 4;           // = the coset size of the last inversion level (k_poly_inverses)
    uint32_t blocks = (uint32_t)((n + (size_t)POLY_THREADS * per_lane - 1) / ((size_t)POLY_THREADS * per_lane));
    if (blocks == 0) blocks = 1;
    if (!d_resident) KZG_HIP_TRY(ctx, set.a.reserve(n * 32));                 // evaluations (wire)
    if (d_resident) evals = nullptr;
    const uint4* d_a = d_resident ? d_resident : set.a.as<uint4>();
    // inverses (planes) | level scratch: the smaller domains' inverses, two ping-pong plane sets of the small kernel
    // The one-workgroup kernel takes the chain up to 2^chain_small_log points, x4 levels on the whole chip go on from there: its late levels
    // keep all 16 waves of one CU busy, a x4 launch over many CUs costs about one of them.  Chain lengths 2..12, off-domain proofs of
    // 2^11 / 2^12 / 2^14 evaluations (tools/time_proof_sizes.py, same box): 12 -> 0.231 / 0.295 / 0.426 ms, 9 -> 0.226 / 0.281 / 0.414, 7 -> 0.230 / 0.286 / 0.422
    constexpr int chain_small_log = 9;
    static_assert(chain_small_log >= 2 && chain_small_log <= (int)POLY_SMALL_MAX_LOG, "the one-workgroup kernel holds the chain's first levels in LDS");
    const size_t chain_small_max = (size_t)1 << chain_small_log;
    const size_t n1 = n > chain_small_max ? n / 4 : 0;       // the level above the last one (0: the small kernel gives all n inverses)
    const size_t lvl_words = n1 ? (n1 + n1 / 2) * NL + 64 : 0;          // sum over n/4, n/16, ... < n1 * 4/3
    KZG_HIP_TRY(ctx, set.b.reserve((n * NL + lvl_words) * 4));
    KZG_HIP_TRY(ctx, set.c.reserve(n * 32));                 // quotient (wire)
    KZG_HIP_TRY(ctx, set.small.reserve(4096 + (size_t)blocks * NL * 4 * 2));
    if (!set.pinned) KZG_HIP_TRY(ctx, hipHostMalloc(&set.pinned, 4096, hipHostMallocDefault));
    uint8_t* small = set.small.as<uint8_t>();
    ProofScalars* ps = reinterpret_cast<ProofScalars*>(small);
    uint4* d_zt = reinterpret_cast<uint4*>(small + 1024);    // zt[a] = z^(2^a), a <= log_n; zt[log_n + 1] = 1 / (1 - z^n); then z^-(2^a) and three constants (on-domain z)
    uint4* d_z = d_zt;
    int32_t* partial = reinterpret_cast<int32_t*>(small + 4096);
    int32_t* d_inv = set.b.as<int32_t>();
    int32_t* d_lvl = d_inv + n * NL;

    uint8_t* pin = static_cast<uint8_t*>(set.pinned);       // [0, 1024): init image, [1024, 3072): the scalars of the inversion chain (zt), [3072, ..): y readback
    ProofScalars* init = reinterpret_cast<ProofScalars*>(pin);
    memset(init, 0, sizeof *init);
    init->on_domain_index = NO_INDEX;
    bool z_on_domain = false;
    {
        uint64_t* zt = reinterpret_cast<uint64_t*>(pin + 1024);
        memcpy(zt, z, 32);
        for (int a = 1; a <= log_n; ++a) h_fr_mul(zt + 4 * (a - 1), zt + 4 * (a - 1), zt + 4 * a);
        const uint64_t one_int[4] = {1, 0, 0, 0};
        uint64_t one_w[4], den[4];
        h_fr_mul(H_FR_R2, one_int, one_w);
        h_fr_sub(one_w, zt + 4 * log_n, den);                // 1 - z^n (zero: z is on the domain, the device inverts what it needs itself)
        uint64_t* top = zt + 4 * (log_n + 1);
        uint64_t* zit = zt + 4 * (log_n + 2);                // z^-(2^a), a <= log_n: only read when z is on the domain
        uint64_t* cst = zt + 4 * (2 * log_n + 3);            // 1/(i - 1), -1/2, 1/(-i - 1), i = w^(n/4) = 5^((r-1)/4)
        if ((den[0] | den[1] | den[2] | den[3]) == 0) {
            z_on_domain = true;
            memset(top, 0, 32);
            h_fr_inv(zt, zit);
            for (int a = 1; a <= log_n; ++a) h_fr_mul(zit + 4 * (a - 1), zit + 4 * (a - 1), zit + 4 * a);
        } else {
            h_fr_inv(den, top);
            memset(zit, 0, (size_t)(log_n + 1) * 32);
        }
        memcpy(cst, h_on_domain_constants(), 96);
    }
    if (!ctx->poly_lds_attr_set) {
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_poly_inv_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(NL * POLY_SMALL_MAX * 4)));
        ctx->poly_lds_attr_set = true;
    }
    // z = w^m on a domain of at most 4 096 points (compute_proof_with_known_z_fr_index at the reference's bench sizes): the host finds m,
    // the inverses come from the domain's table 1 / (w^k - 1) -- no inversion chain, no barycentric sum
    uint32_t m_known = NO_INDEX;
    if (want_proof && z_on_domain && n >= 2 && n <= POLY_SMALL_MAX && h_domain_index(z, log_n, &m_known)) {
        int32_t*& t1 = ctx->ondomain_inv[log_n];
        if (!t1) {                                           // once per domain size: the chain below at z = 1
            std::vector<uint64_t> z1((size_t)(2 * log_n + 6) * 4, 0);
            const uint64_t one_int[4] = {1, 0, 0, 0};
            uint64_t one_w[4];
            h_fr_mul(H_FR_R2, one_int, one_w);
            for (int a = 0; a <= log_n; ++a) { memcpy(&z1[4 * (size_t)a], one_w, 32); memcpy(&z1[4 * (size_t)(log_n + 2 + a)], one_w, 32); }
            memcpy(&z1[4 * (size_t)(2 * log_n + 3)], h_on_domain_constants(), 96);
            uint4* d_z1 = nullptr;
            KZG_HIP_TRY(ctx, hipMalloc(&d_z1, z1.size() * 8));
            hipError_t e1 = hipMalloc(&t1, (size_t)NL * n * 4);
            if (e1 == hipSuccess) e1 = hipMemcpy(d_z1, z1.data(), z1.size() * 8, hipMemcpyHostToDevice);
            if (e1 == hipSuccess) {
                hipLaunchKernelGGL(k_poly_inv_small, dim3(1), dim3(POLY_SMALL_THREADS), (size_t)NL * n * 4, st, d_z1, log_n, log_n, tb, t1);
                e1 = hipGetLastError();
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
            }
            (void)hipFree(d_z1);
            if (e1 != hipSuccess) { if (t1) { (void)hipFree(t1); t1 = nullptr; } KZG_HIP_TRY(ctx, e1); }
        }
        // (no upload of the ProofScalars image and no copy back: the kernels below only WRITE it, straight into the pinned read-back slot)
        void* ps_host_dev = nullptr;
        KZG_HIP_TRY(ctx, hipHostGetDevicePointer(&ps_host_dev, pin + 3072, 0));
        ProofScalars* ps_out = static_cast<ProofScalars*>(ps_host_dev);
        if (evals) KZG_HIP_TRY(ctx, hipMemcpyAsync(set.a.p, evals, n * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_poly_quotient_table, dim3(blocks), dim3(POLY_THREADS), 0, st, d_a, (uint32_t)n, tb, t1, m_known,
                           set.c.as<uint4>(), partial, ps_out);
        if (blocks > 1)
            hipLaunchKernelGGL(k_poly_quotient_on_domain_known, dim3(1), dim3(POLY_THREADS), 0, st, d_a, (uint32_t)n, tb, partial, blocks,
                               m_known, ps_out, set.c.as<uint4>());
        KZG_HIP_TRY(ctx, hipGetLastError());
        return skip_intt ? KZG_OK : ntt_run(ctx, set.c.p, n, true, st, nttws);
    }
    // one upload for the scalar image (pin[0, 1024) -> small[0, 1024)) and the chain's scalars right behind it (pin + 1024 -> small + 1024)
    static_assert(sizeof(ProofScalars) <= 1024, "the ProofScalars image and the zt table are uploaded as one block");
    memset(pin + sizeof(ProofScalars), 0, 1024 - sizeof(ProofScalars));
    // The chain below needs z only, not the evaluations: for a proof from HOST evaluations it is enqueued FIRST, on the context's auxiliary stream, and
    // runs while this thread is inside the pageable upload of the evaluations (0.6 ms at 2^20; the chain: six launches, ~0.09 ms of latency).
    hipStream_t sc = st;
    if (evals && n > chain_small_max) {
        int32_t rca = ctx_aux_stream(ctx, st, &sc);
        if (rca != KZG_OK) return rca;
    }
    KZG_HIP_TRY(ctx, hipMemcpyAsync(small, pin, 1024 + (size_t)(2 * log_n + 6) * 32, hipMemcpyHostToDevice, sc));
    if (evals && sc == st) KZG_HIP_TRY(ctx, hipMemcpyAsync(set.a.p, evals, n * 32, hipMemcpyHostToDevice, st));   // nullptr: set.a already holds the n evaluations (blob proofs)

    // the chain of smaller domains, coarsest first: small kernel (<= 4096 points, in LDS), then x4 levels, then the last level
    const int32_t* next = nullptr;
    int direct = 0;
    if (n <= chain_small_max) {
        hipLaunchKernelGGL(k_poly_inv_small, dim3(1), dim3(POLY_SMALL_THREADS), (size_t)NL * n * 4, st, d_zt, log_n, log_n, tb, d_inv);
        next = d_inv; direct = 1;
    } else {
        int log_l = log_n - 2;                               // sizes n/4, n/16, .. down to the first one <= 4096
        std::vector<int> logs;
        while (log_l > chain_small_log) { logs.push_back(log_l); log_l -= 2; }
        std::vector<int32_t*> bufs;                          // level buffers inside d_lvl: size 2^logs[0] first
        int32_t* cursor = d_lvl;
        for (int l : logs) { bufs.push_back(cursor); cursor += ((size_t)NL << l); }
        int32_t* small_out = cursor;                         // 2^log_l <= 4096 entries
        hipLaunchKernelGGL(k_poly_inv_small, dim3(1), dim3(POLY_SMALL_THREADS), ((size_t)NL << log_l) * 4, sc, d_zt, log_n, log_l, tb, small_out);
        const int32_t* prev = small_out;
        for (int q = (int)logs.size() - 1; q >= 0; --q) {
            const uint32_t T = 1u << (logs[q] - 2);
            hipLaunchKernelGGL(k_poly_inv_level, dim3((T + POLY_THREADS - 1) / POLY_THREADS), dim3(POLY_THREADS), 0, sc, d_zt, log_n, logs[q], tb,
                               prev, bufs[q]);
            prev = bufs[q];
        }
        next = prev;
        if (sc != st) {                                      // the upload of the evaluations now (the host sits in it while the chain runs), then join
            KZG_HIP_TRY(ctx, hipGetLastError());
            if (!set.ev_chain) KZG_HIP_TRY(ctx, hipEventCreateWithFlags(&set.ev_chain, hipEventDisableTiming));
            KZG_HIP_TRY(ctx, hipEventRecord(set.ev_chain, sc));
            KZG_HIP_TRY(ctx, hipMemcpyAsync(set.a.p, evals, n * 32, hipMemcpyHostToDevice, st));
            KZG_HIP_TRY(ctx, hipStreamWaitEvent(st, set.ev_chain, 0));
        }
    }
    const int fused_y = blocks == 1 && !z_on_domain;         // one workgroup holds the whole barycentric sum: no second launch for y
    hipLaunchKernelGGL(k_poly_inverses, dim3(blocks), dim3(POLY_THREADS), 0, st, d_a, (uint32_t)n, log_n, tb, d_z,
                       next, direct, d_inv, partial, ps, fused_y);
    if (!fused_y)
        hipLaunchKernelGGL(k_poly_finish_y, dim3(1), dim3(POLY_THREADS), 0, st, d_a, (uint32_t)n, log_n, d_z,
                           partial, blocks, ps);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(pin + 3072, ps, sizeof(ProofScalars), hipMemcpyDeviceToHost, st));
    if (!want_proof) return KZG_OK;
    hipLaunchKernelGGL(k_poly_quotient, dim3(blocks), dim3(POLY_THREADS), 0, st, d_a, (uint32_t)n, tb, d_inv,
                       ps, set.c.as<uint4>(), partial);
    if (z_on_domain)                                         // (z off the domain: the kernel would return at once)
        hipLaunchKernelGGL(k_poly_quotient_on_domain, dim3(1), dim3(POLY_THREADS), 0, st, (uint32_t)n, tb, partial, blocks, ps,
                           set.c.as<uint4>());
    KZG_HIP_TRY(ctx, hipGetLastError());
    // commit_eval_form(quotient): coefficients = IFFT(q), then MSM over the monomial SRS (kzg.rs:176-177)
    return skip_intt ? KZG_OK : ntt_run(ctx, set.c.p, n, true, st, nttws);
}
// The cached Lagrange basis of exactly n points, if the SRS carries one (without one: the IFFT + monomial-basis form)
static const kzg_srs* proof_lagrange_basis(const kzg_srs* srs, size_t n) {
    if (n < 2 || srs->lagrange_of != 0) return nullptr;
    return srs_cached_lagrange(srs, n);
}
static void proof_read_y(const PolySet& set, uint64_t* out_y) {
    const ProofScalars* host = reinterpret_cast<const ProofScalars*>(static_cast<const uint8_t*>(set.pinned) + 3072);
    memcpy(out_y, host->y_wire, 32);
}

int32_t proof_run(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4],
                  uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y, bool want_proof, size_t coeff_lo, uint64_t* out_xyzz) {
    if (ctx->slot_pending[0]) {
        ctx->last_error = "a kzg_*_begin on slot 0 is still in flight: call its end first";
        return KZG_ERR_INVALID_ARG;
    }
    PolySet& set = ctx->poly[0];
    hipStream_t st = ctx->stream;
    // a Lagrange basis of n points cached with the SRS (kzg_srs_cache_lagrange): the quotient is committed in evaluation form, no IFFT
    const kzg_srs* lag = (want_proof && srs && coeff_lo == 0) ? proof_lagrange_basis(srs, n) : nullptr;
    int32_t rc = proof_enqueue(ctx, set, st, &ctx->ntt, evals, n, z, want_proof, lag != nullptr);
    if (rc != KZG_OK) { (void)hipStreamSynchronize(st); return rc; }
    if (lag) {
        rc = msm_run(ctx, srs_bases(lag, 0, n, ctx->msm_c_override == 0), set.c.p, n, out_xy, out_inf, out_xyzz);
        if (rc == KZG_OK && out_y) proof_read_y(set, out_y);
        return rc;
    }
    // the whole SRS commits the whole quotient; a shard holding powers [coeff_lo, coeff_lo + srs->n) commits its slice of it
    if (!want_proof || coeff_lo >= n) {
        KZG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (out_y) proof_read_y(set, out_y);
        if (want_proof) {
            if (out_xyzz) memset(out_xyzz, 0, 128);
            if (out_xy) { memset(out_xy, 0, 64); if (out_inf) *out_inf = 1; }
        }
        return KZG_OK;
    }
    const size_t len = std::min(srs->n, n - coeff_lo);
    rc = msm_run(ctx, srs_bases(srs, 0, len, ctx->msm_c_override == 0), set.c.as<uint4>() + 2 * coeff_lo, len, out_xy, out_inf, out_xyzz);
    if (rc == KZG_OK && out_y) proof_read_y(set, out_y);      // msm_run has synchronised the stream
    return rc;
}

// asynchronous form: everything on the slot's stream; proof_end collects the point and y
int32_t proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4], int slot, const void* d_resident) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    PolySet& set = ctx->poly[slot];
    const kzg_srs* lag = proof_lagrange_basis(srs, n);
    rc = proof_enqueue(ctx, set, st, &ctx->slot_ntt(slot), evals, n, z, true, lag != nullptr, static_cast<const uint4*>(d_resident));
    if (rc != KZG_OK) { (void)hipStreamSynchronize(st); return rc; }
    return msm_begin(ctx, slot, srs_bases(lag ? lag : srs, 0, n, ctx->msm_c_override == 0), set.c.p, n);
}
int32_t proof_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y) {
    int32_t rc = msm_end(ctx, slot, out_xy, out_inf, nullptr);
    if (rc == KZG_OK && out_y) proof_read_y(ctx->poly[slot], out_y);
    return rc;
}

// Batched y_i = p_i(z_i) for `nb` blobs whose bytes sit packed in PINNED host memory `packed` (blob i at meta[i].off, 32-byte aligned,
// zero-filled up to the next 32-byte chunk) with the challenges zs (wire).  Three steps so that the caller can overlap the uploads with
// the hashing of later blobs:
//   vb_evaluate_setup(ctx, packed_len, nb)                     buffers, the 4096-point domain table, kernel attribute
//   vb_evaluate_enqueue(ctx, packed, meta, b0, b1, zs)         blobs [b0, b1): their bytes, challenges and descriptors go up and the two
//                                                              kernels are enqueued on the context's stream; returns at once; may be called
//                                                              from any host thread, chunks in any order (they touch disjoint ranges)
//   vb_evaluate_finish(ctx, nb, ys_out, fallback_out)          waits; ys_out = nb wire elements; fallback_out[i] != 0 marks the blobs this
//                                                              path does not cover (z on the domain, more than 2^VB_MAX_LOG elements)
// Slot 0's buffers and stream.
int32_t vb_evaluate_setup(kzg_ctx* ctx, size_t packed_len, size_t nb) {
    static_assert(sizeof(VbBlob) == 16, "VbBlob layout is part of the host interface (capi.hip)");
    PolySet& set = ctx->poly[0];
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, VB_MAX_LOG, false, &tb);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, set.c.reserve(packed_len + 64));
    KZG_HIP_TRY(ctx, set.a.reserve(nb * 32 * 2 + 64));                                 // zs | ys
    KZG_HIP_TRY(ctx, set.b.reserve(nb * (sizeof(VbBlob) + sizeof(VbPrep)) + 64));
    KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_vb_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NL * 2 * VB_THREADS * 4)));
    KZG_HIP_TRY(ctx, hipMemsetAsync(set.a.as<uint4>() + 2 * nb, 0, nb * 32, ctx->stream));
    // slot 1's stream carries the blob bytes: created HERE, on the calling thread under ctx->mu -- vb_evaluate_enqueue runs on the
    // caller's worker threads, which must not create context state (two chunks finishing together raced on stream_x[0])
    hipStream_t st_copy = nullptr;
    return msm_slot_stream(ctx, 1, &st_copy);
}
// `small_pinned`: pinned staging of at least nb x 48 bytes for the chunk's challenges and descriptors (a copy from pageable memory would
// make this call wait for everything queued on the stream before it)
int32_t vb_evaluate_enqueue(kzg_ctx* ctx, const uint8_t* packed, const void* meta_host, size_t nb, size_t b0, size_t b1, const uint64_t* zs,
                            uint8_t* small_pinned) {
    if (b1 <= b0) return KZG_OK;
    // worker threads of the caller's pool: errors travel in the return code only (ctx->last_error belongs to the thread holding ctx->mu)
    hipEvent_t up = nullptr;
#define VB_TRY(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); if (up) (void)hipEventDestroy(up); return KZG_ERR_DEVICE; } } while (0)
    VB_TRY(hipSetDevice(ctx->device));
    PolySet& set = ctx->poly[0];
    hipStream_t st = ctx->stream, st_copy = ctx->stream_x[0] ? ctx->stream_x[0] : ctx->stream;   // slot 1's stream exists since vb_evaluate_setup
    // the blob bytes go up on a second stream (slot 1's): k_vb_prep is one inversion deep (~0.2 ms on a few lone waves, whatever the
    // chunk size) and needs the challenges only, so it runs while the chunk's bytes are still on the bus; k_vb_eval waits for them
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, VB_MAX_LOG, false, &tb);                          // cached by vb_evaluate_setup: a look-up
    if (rc != KZG_OK) return rc;
    const VbBlob* meta = static_cast<const VbBlob*>(meta_host);
    uint4* d_zs = set.a.as<uint4>();
    uint4* d_ys = d_zs + 2 * nb;
    VbBlob* d_meta = set.b.as<VbBlob>();
    VbPrep* d_prep = reinterpret_cast<VbPrep*>(d_meta + nb);
    uint32_t max_log = 0;
    size_t lo = SIZE_MAX, hi = 0;
    for (size_t i = b0; i < b1; ++i) {
        if (meta[i].log_n > (uint32_t)VB_MAX_LOG) continue;
        max_log = std::max(max_log, meta[i].log_n);
        lo = std::min<size_t>(lo, meta[i].off);
        hi = std::max<size_t>(hi, meta[i].off + ((size_t)meta[i].len + 31) / 32 * 32);
    }
    const uint32_t max_lf = std::min<uint32_t>(1u << max_log, (uint32_t)VB_THREADS);
    const size_t lds = std::max<size_t>((size_t)NL * 2 * max_lf, (size_t)NL * VB_THREADS) * 4;
    if (hi > lo) {
        VB_TRY(hipMemcpyAsync(set.c.as<uint8_t>() + lo, packed + lo, hi - lo, hipMemcpyHostToDevice, st_copy));
        VB_TRY(hipEventCreateWithFlags(&up, hipEventDisableTiming));
        VB_TRY(hipEventRecord(up, st_copy));
    }
    uint8_t* pz = small_pinned + b0 * 32;
    uint8_t* pm = small_pinned + nb * 32 + b0 * sizeof(VbBlob);
    memcpy(pz, zs + 4 * b0, (b1 - b0) * 32);
    memcpy(pm, meta + b0, (b1 - b0) * sizeof(VbBlob));
    VB_TRY(hipMemcpyAsync(d_zs + 2 * b0, pz, (b1 - b0) * 32, hipMemcpyHostToDevice, st));
    VB_TRY(hipMemcpyAsync(d_meta + b0, pm, (b1 - b0) * sizeof(VbBlob), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_vb_prep, dim3((unsigned)((b1 - b0 + 63) / 64)), dim3(64), 0, st, d_zs + 2 * b0, d_meta + b0, (uint32_t)(b1 - b0), d_prep + b0);
    if (up) {
        VB_TRY(hipStreamWaitEvent(st, up, 0));
        (void)hipEventDestroy(up);                                                     // released by the runtime once the wait has been satisfied
    }
    hipLaunchKernelGGL(k_vb_eval, dim3((unsigned)(b1 - b0)), dim3(VB_THREADS), lds, st, set.c.as<uint8_t>(), d_meta + b0, d_prep + b0, tb, d_ys + 2 * b0);
    VB_TRY(hipGetLastError());
#undef VB_TRY
    return KZG_OK;
}
int32_t vb_evaluate_finish(kzg_ctx* ctx, size_t nb, uint64_t* ys_out, uint8_t* fallback_out) {
    PolySet& set = ctx->poly[0];
    hipStream_t st = ctx->stream;
    uint4* d_ys = set.a.as<uint4>() + 2 * nb;
    VbPrep* d_prep = reinterpret_cast<VbPrep*>(set.b.as<VbBlob>() + nb);
    static thread_local std::vector<VbPrep> prep_host;
    prep_host.resize(nb);
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ys_out, d_ys, nb * 32, hipMemcpyDeviceToHost, st));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(prep_host.data(), d_prep, nb * sizeof(VbPrep), hipMemcpyDeviceToHost, st));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < nb; ++i) fallback_out[i] = prep_host[i].fallback ? 1 : 0;
    return KZG_OK;
}

}  // namespace kzg
