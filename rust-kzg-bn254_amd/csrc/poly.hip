// poly.hip — the O(n) polynomial work of KZG::compute_proof_impl (prover/src/kzg.rs:128-178) on the GPU:
//   y   = p(z)                    barycentric, primitives/src/helpers.rs:475-535 (incl. the z-on-domain early return :497-504)
//   q_i = (f_i - y) / (w^i - z)   kzg.rs:151-174
//   q_m = sum_{i != m} (f_i - y) w^i / (z (z - w^i))   when z = w^m, kzg.rs:237-260
// then the quotient is committed as MSM(srs, IFFT(q)) (see kzg_commit_eval_form).
// The reference performs one field inversion per division (2n-3n serial inversions); here all
// denominators w^i - z share one batch inversion (Montgomery's trick: per-lane prefix products, one
// Fermat inversion per lane, back-substitution).  Results are identical field elements.
// Also: helpers::calculate_roots_of_unity (helpers.rs:553-589) as a kernel.
#include "engine.h"
#include "field29.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kzg {

constexpr int POLY_THREADS = 256;
constexpr uint32_t NO_INDEX = 0xFFFFFFFFu;

__device__ __forceinline__ void pl_load(Fr& v, const int32_t* __restrict__ planes, size_t stride, size_t i) {
#pragma unroll
    for (int j = 0; j < NL; ++j) v.l[j] = planes[(size_t)j * stride + i];
}
__device__ __forceinline__ void pl_store(int32_t* __restrict__ planes, size_t stride, size_t i, const Fr& v) {
#pragma unroll
    for (int j = 0; j < NL; ++j) planes[(size_t)j * stride + i] = v.l[j];
}
__device__ __forceinline__ void domain_elem(Fr& w, const NttTables& tb, uint32_t E) {   // w^E, result in (-m, 2m)
    pl_load(w, tb.lo, tb.lo_len, E & (tb.lo_len - 1));
    uint32_t eh = E >> tb.lo_bits;
    if (eh != 0) {
        Fr h;
        pl_load(h, tb.hi, tb.hi_len, eh);
        fe_mul(w, w, h);
    }
}
__device__ __forceinline__ void wire_load(Fr& v, const uint4* __restrict__ src, size_t i) {   // wire -> internal, (-m, 2m)
    uint4 a = src[2 * i], b = src[2 * i + 1];
    uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fe_from_wire(v, w32);
}
__device__ __forceinline__ void wire_store(uint4* __restrict__ dst, size_t i, const Fr& v) {  // internal (|v| < 169 m) -> wire
    uint32_t w32[8];
    fe_to_wire(w32, v);
    dst[2 * i] = make_uint4(w32[0], w32[1], w32[2], w32[3]);
    dst[2 * i + 1] = make_uint4(w32[4], w32[5], w32[6], w32[7]);
}
// a^(r-2) by square-and-multiply over the modulus words (Fermat)
__device__ __noinline__ void fr_inverse(Fr& out, const Fr& a) {
    Fr acc, base = a;
    fe_set_one(acc);
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = FrParams::P32[j];
    e[0] -= 2u;                                   // r - 2 (no borrow: low word of r is 0xf0000001)
    for (int w = 0; w < 8; ++w) {
        uint32_t bits = e[w];
        for (int b = 0; b < 32; ++b) {
            if (w == 7 && b >= 30) break;        // r < 2^254
            if (bits & 1u) fe_mul(acc, acc, base);
            fe_sqr(base, base);
            bits >>= 1;
        }
    }
    out = acc;
}
// block-wide sum of one Fr per thread (values in (-m, 2m)); result (reduced) valid in thread 0
__device__ __forceinline__ void block_sum(Fr& v, int32_t* lds /* NL * POLY_THREADS */) {
    const int t = threadIdx.x;
    int level = 0;
    for (int d = POLY_THREADS / 2; d >= 1; d >>= 1, ++level) {
#pragma unroll
        for (int j = 0; j < NL; ++j) lds[j * POLY_THREADS + t] = v.l[j];
        __syncthreads();
        if (t < d) {
            Fr u;
#pragma unroll
            for (int j = 0; j < NL; ++j) u.l[j] = lds[j * POLY_THREADS + t + d];
            fe_add(v, v, u);
            fe_norm(v);
            if (level == 3) fe_reduce(v);        // 16 terms so far: |v| < 32 m -> back to (-m, 2m)
        }
        __syncthreads();
    }
    if (t == 0) fe_reduce(v);
}

struct ProofScalars {        // device-resident small state of one proof computation
    uint32_t on_domain_index; // NO_INDEX if z is not a domain element
    uint32_t pad[3];
    int32_t y[NL];            // y = p(z), internal form, reduced
    uint32_t y_wire[8];
};

// ---- K1: denominators, batch inversion, barycentric partial sums -------------------------------------
// lane t owns elements i = k * T + t.  inv[i] = 1 / (w^i - z)  (1 for the on-domain index).
__global__ void __launch_bounds__(POLY_THREADS)
k_poly_inverses(const uint4* __restrict__ evals, uint32_t n, NttTables tb, const uint4* __restrict__ z_wire,
                int32_t* __restrict__ inv, int32_t* __restrict__ dvals, int32_t* __restrict__ partial /* NL x gridDim */,
                ProofScalars* __restrict__ ps) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr z;
    wire_load(z, z_wire, 0);
    fe_canon(z);
    Fr run;
    fe_set_one(run);
    // forward: prefix products
    for (uint32_t i = t; i < n; i += T) {
        Fr w, d;
        domain_elem(w, tb, i);
        fe_canon(w);
        fe_sub(d, w, z);                          // canonical - canonical: in (-m, m), limbs within +-2^29
        bool zero = fe_is_literal_zero(d);
        if (zero) { ps->on_domain_index = i; fe_set_one(d); }
        pl_store(dvals, n, i, d);
        pl_store(inv, n, i, run);                 // product of this lane's earlier denominators
        fe_mul(run, run, d);
    }
    Fr rinv;
    fr_inverse(rinv, run);
    // backward: inv_i = rinv * prefix_i ; rinv *= d_i
    const uint32_t cnt = (n > t) ? (n - 1 - t) / T + 1 : 0;
    Fr sum;
    fe_set_zero(sum);
    for (uint32_t k = cnt; k-- > 0;) {
        const uint32_t i = t + k * T;
        Fr pre, d, iv;
        pl_load(pre, inv, n, i);
        pl_load(d, dvals, n, i);
        fe_mul(iv, rinv, pre);
        fe_mul(rinv, rinv, d);
        pl_store(inv, n, i, iv);
        // barycentric term f_i w^i / (z - w^i) = -(f_i w^i inv_i)
        Fr f, w, term;
        wire_load(f, evals, i);
        domain_elem(w, tb, i);
        fe_mul(term, f, w);
        fe_mul(term, term, iv);
        fe_sub(sum, sum, term);
        fe_norm(sum);                             // |sum| grows by 2m per term
        if ((k & 31u) == 0) fe_reduce(sum);
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x == 0) pl_store(partial, gridDim.x, blockIdx.x, sum);
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
        Fr z, zn, one, ninv;
        wire_load(z, z_wire, 0);
        zn = z;
        for (int k = 0; k < log_n; ++k) fe_sqr(zn, zn);      // z^n, n = 2^log_n
        fe_set_one(one);
        fe_sub(zn, zn, one);                       // in (-3m, 2m)
#pragma unroll
        for (int j = 0; j < NL; ++j) ninv.l[j] = (int32_t)FrParams::NINV[log_n * NL + j];
        fe_mul(y, sum, zn);
        fe_mul(y, y, ninv);                        // helpers.rs:529-532
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

// ---- blob bytes -> Fr (helpers::to_fr_array, primitives/src/helpers.rs:40-57) ---------------------------------------
// element i = the 32 big-endian bytes [32 i, 32 i + 32) (the last chunk right-padded with zeros) mod r, emitted in wire
// (Montgomery, radix 2^256) form; elements i >= n_elems (power-of-two padding of PolynomialEvalForm::new,
// polynomial.rs:49-51) are zero.  One multiply: x * (2^(256+261) mod r) * 2^-261 = x * 2^256 mod r, x < 2^256 < 5.3 r.
__global__ void __launch_bounds__(POLY_THREADS)
k_blob_to_fr(const uint8_t* __restrict__ bytes, size_t len, uint32_t n_elems, uint32_t n_padded, uint4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    uint32_t w32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n_elems) {
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
static int32_t proof_enqueue(kzg_ctx* ctx, PolySet& set, hipStream_t st, NttWorkspace* nttws, const uint64_t* evals, size_t n,
                             const uint64_t z[4], bool want_proof) {
    int log_n = ilog2_exact(n);
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, log_n, false, &tb);
    if (rc != KZG_OK) return rc;
    if (want_proof && n > 1) { NttTables tbi; rc = ntt_get_tables(ctx, log_n, true, &tbi); if (rc != KZG_OK) return rc; }
    // lanes: about `per_lane` elements each, at least one block.  Every lane pays one Fermat inversion (~380 dependent multiplies,
    // the latency floor of this kernel), so fewer elements per lane shorten the serial part until the extra waves cost more.
    static int per_lane = 0;
    if (per_lane == 0) { const char* env = getenv("KZG_POLY_PER_LANE"); per_lane = env && atoi(env) > 0 ? atoi(env) : 16; }   // measured at 2^20: 32 -> 3.42 ms, 16 -> 3.32 ms, 8 -> 3.41 ms per proof
    uint32_t blocks = (uint32_t)((n + (size_t)POLY_THREADS * per_lane - 1) / ((size_t)POLY_THREADS * per_lane));
    if (blocks == 0) blocks = 1;
    KZG_HIP_TRY(ctx, set.a.reserve(n * 32));                 // evaluations (wire)
    KZG_HIP_TRY(ctx, set.b.reserve(n * NL * 4 * 2));         // inverses | denominators (planes)
    KZG_HIP_TRY(ctx, set.c.reserve(n * 32));                 // quotient (wire)
    KZG_HIP_TRY(ctx, set.small.reserve(4096 + (size_t)blocks * NL * 4 * 2));
    if (!set.pinned) KZG_HIP_TRY(ctx, hipHostMalloc(&set.pinned, 4096, hipHostMallocDefault));
    uint8_t* small = set.small.as<uint8_t>();
    ProofScalars* ps = reinterpret_cast<ProofScalars*>(small);
    uint4* d_z = reinterpret_cast<uint4*>(small + 1024);
    int32_t* partial = reinterpret_cast<int32_t*>(small + 4096);
    int32_t* d_inv = set.b.as<int32_t>();
    int32_t* d_den = d_inv + n * NL;

    uint8_t* pin = static_cast<uint8_t*>(set.pinned);       // [0, 1024): init image, [1024, 1056): z, [2048, ..): y readback
    ProofScalars* init = reinterpret_cast<ProofScalars*>(pin);
    memset(init, 0, sizeof *init);
    init->on_domain_index = NO_INDEX;
    memcpy(pin + 1024, z, 32);
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ps, init, sizeof *init, hipMemcpyHostToDevice, st));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(d_z, pin + 1024, 32, hipMemcpyHostToDevice, st));
    if (evals) KZG_HIP_TRY(ctx, hipMemcpyAsync(set.a.p, evals, n * 32, hipMemcpyHostToDevice, st));   // nullptr: set.a already holds the n evaluations (blob proofs)

    hipLaunchKernelGGL(k_poly_inverses, dim3(blocks), dim3(POLY_THREADS), 0, st, set.a.as<uint4>(), (uint32_t)n, tb, d_z,
                       d_inv, d_den, partial, ps);
    hipLaunchKernelGGL(k_poly_finish_y, dim3(1), dim3(POLY_THREADS), 0, st, set.a.as<uint4>(), (uint32_t)n, log_n, d_z,
                       partial, blocks, ps);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(pin + 2048, ps, sizeof(ProofScalars), hipMemcpyDeviceToHost, st));
    if (!want_proof) return KZG_OK;
    hipLaunchKernelGGL(k_poly_quotient, dim3(blocks), dim3(POLY_THREADS), 0, st, set.a.as<uint4>(), (uint32_t)n, tb, d_inv,
                       ps, set.c.as<uint4>(), partial);
    hipLaunchKernelGGL(k_poly_quotient_on_domain, dim3(1), dim3(POLY_THREADS), 0, st, (uint32_t)n, tb, partial, blocks, ps,
                       set.c.as<uint4>());
    KZG_HIP_TRY(ctx, hipGetLastError());
    // commit_eval_form(quotient): coefficients = IFFT(q), then MSM over the monomial SRS (kzg.rs:176-177)
    return ntt_run(ctx, set.c.p, n, true, st, nttws);
}
static void proof_read_y(const PolySet& set, uint64_t* out_y) {
    const ProofScalars* host = reinterpret_cast<const ProofScalars*>(static_cast<const uint8_t*>(set.pinned) + 2048);
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
    int32_t rc = proof_enqueue(ctx, set, st, &ctx->ntt, evals, n, z, want_proof);
    if (rc != KZG_OK) { (void)hipStreamSynchronize(st); return rc; }
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
int32_t proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4], int slot) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    PolySet& set = ctx->poly[slot];
    rc = proof_enqueue(ctx, set, st, &ctx->slot_ntt(slot), evals, n, z, true);
    if (rc != KZG_OK) { (void)hipStreamSynchronize(st); return rc; }
    return msm_begin(ctx, slot, srs_bases(srs, 0, n, ctx->msm_c_override == 0), set.c.p, n);
}
int32_t proof_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y) {
    int32_t rc = msm_end(ctx, slot, out_xy, out_inf, nullptr);
    if (rc == KZG_OK && out_y) proof_read_y(ctx->poly[slot], out_y);
    return rc;
}

}  // namespace kzg
