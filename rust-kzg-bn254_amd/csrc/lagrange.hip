// lagrange.hip — BASELINE config 4 sharded by EVALUATION index: commitments and proofs from a SLICE of the evaluations and the matching
// slice of the Lagrange basis L_i = g1_ifft(srs)[i] (prover/src/kzg.rs:263-285).
//
// The reference commits an evaluation-form polynomial as MSM(g1_ifft(srs), evals) (kzg.rs:96-100) and a proof as the same MSM over the
// evaluations of the quotient (kzg.rs:176-177, :151-174, on-domain entry :237-260).  Both are sums over the evaluation index i:
//     C  = sum_i f_i L_i                                   -> rank g: sum over its slice, ONE exchange of the partial points
//     y  = (z^n - 1) / n  sum_i f_i w^i / (z - w^i)        -> rank g: partial sum S_g, exchange of G x 32 B (helpers.rs:507-532)
//     q_i = (f_i - y) / (w^i - z),  pi = sum_i q_i L_i     -> pointwise on the slice, then the slice's MSM, exchange of the partial points
//     z = w^m:  y = f_m (helpers.rs:497-504),  q_m = -(1/z) sum_{i != m} q_i w^i (kzg.rs:237-260)  -> a second partial sum T_g that
//                                                         rides the exchange of the points; q_m L_m is added by the fold
// so a rank uploads, inverts and divides ITS slice only: no replicated upload, no IFFT, no whole-polynomial quotient (round 4's
// kzg_*_partial forms did all three on every rank).
//
// Denominators of a contiguous slice have no closed-form product (poly.hip's coset recursion needs the whole domain), so a slice is
// inverted with Montgomery's trick per workgroup of 1 024 elements: lane products -> product tree in LDS -> ONE Bernstein-Yang inversion
// per workgroup (fe_invert.h) -> back down the tree -> the lane's four inverses.  The workgroups run side by side, so a slice costs one
// inversion of latency whatever its length.
#include "poly_common.h"
#include "fe_invert.h"
#include "host_lagrange.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace kzg {

constexpr int LAG_PER = 4;                                  // elements per lane
constexpr uint32_t LAG_BLOCK = POLY_THREADS * LAG_PER;      // elements per workgroup = per inversion

// ---- L1: inverses of w^(base + i) - z over the slice + the slice's part of the barycentric sum ------------------------------------
// lane t of workgroup b owns the slice elements i = 1024 b + t + 256 k, k < 4.  inv (limb planes, stride len) receives 1 / (w^(base+i) - z),
// and 1 for the element with w^(base+i) = z (whose slice index goes to *on_domain).  partial[b] = -sum f_i w^i inv_i = sum f_i w^i / (z - w^i).
__global__ void __launch_bounds__(POLY_THREADS)
k_lag_inverses(const uint4* __restrict__ evals, uint32_t len, uint32_t base, NttTables tb, const uint4* __restrict__ z_wire,
               int32_t* __restrict__ inv, int32_t* __restrict__ partial /* NL x gridDim */, uint32_t* __restrict__ on_domain) {
    __shared__ int32_t tree[NL * 2 * POLY_THREADS];          // heap order: root 1, leaves POLY_THREADS + t
    constexpr uint32_t Lf = POLY_THREADS, S = 2 * POLY_THREADS;
    const uint32_t t = threadIdx.x, i0 = blockIdx.x * LAG_BLOCK + t;
    Fr z;
    wire_load(z, z_wire, 0);
    fe_canon(z);
    Fr d[LAG_PER], pre[LAG_PER], p;
#pragma unroll
    for (int k = 0; k < LAG_PER; ++k) {
        const uint32_t i = i0 + (uint32_t)k * POLY_THREADS;
        if (i < len) {
            Fr w;
            domain_elem(w, tb, base + i);
            fe_canon(w);
            fe_sub(d[k], w, z);                              // canonical - canonical: limbs within +-2^29, |d| < m
            if (fe_is_literal_zero(d[k])) { *on_domain = i; fe_set_one(d[k]); }
        } else {
            fe_set_one(d[k]);
        }
        if (k == 0) p = d[0];
        else { pre[k] = p; fe_mul(p, p, d[k]); }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) tree[j * S + Lf + t] = p.l[j];
    __syncthreads();
    for (uint32_t s = Lf >> 1; s >= 1; s >>= 1) {            // up-sweep: node = product of its two children
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
    if (t == 0) {                                            // the workgroup's one inversion (no denominator is zero: replaced by 1 above)
        Fr root, ri;
#pragma unroll
        for (int j = 0; j < NL; ++j) root.l[j] = tree[j * S + 1];
#ifdef KZG_PROBE_NO_LAG_INVERSION      // measurement build only (tools/build_variant.sh): the inversion priced at ZERO -- WRONG results; bounds what any faster inversion could gain
        ri = root;
#else
        fe_inverse_safegcd(ri, root);
#endif
#pragma unroll
        for (int j = 0; j < NL; ++j) tree[j * S + 1] = ri.l[j];
    }
    __syncthreads();
    for (uint32_t s = 1; s < Lf; s <<= 1) {                  // down-sweep: inverse of a child = inverse of the node x its sibling
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
    Fr inv_all, sum;
#pragma unroll
    for (int j = 0; j < NL; ++j) inv_all.l[j] = tree[j * S + Lf + t];
    fe_set_zero(sum);
#pragma unroll
    for (int k = LAG_PER - 1; k >= 0; --k) {
        Fr iv;
        if (k > 0) { fe_mul(iv, inv_all, pre[k]); fe_mul(inv_all, inv_all, d[k]); }
        else iv = inv_all;
        const uint32_t i = i0 + (uint32_t)k * POLY_THREADS;
        if (i < len) {
            pl_store(inv, len, i, iv);
            if (!evals) continue;                            // inverses only: the sum is k_lag_bary's (uniform across the grid)
            Fr f, w, term;
            wire_load(f, evals, i);
            domain_elem(w, tb, base + i);
            fe_mul(term, f, w);
            fe_mul(term, term, iv);
            fe_sub(sum, sum, term);                          // f_i w^i / (z - w^i) = -(f_i w^i inv_i)
            fe_norm(sum);                                    // four terms of (-m, 2m)
        }
    }
    if (!evals) return;
    fe_reduce(sum);
    __syncthreads();                                         // the tree is dead: its first planes carry the workgroup sum
    block_sum(sum, tree);
    if (t == 0) pl_store(partial, gridDim.x, blockIdx.x, sum);
}

// ---- L1b: the slice's part of the barycentric sum from inverses that are already there (evaluations from a HOST buffer: the inverses need z
// only, so k_lag_inverses(evals = nullptr) runs BESIDE the upload and this kernel behind both).  Same element-to-lane map as L1. ---------------
__global__ void __launch_bounds__(POLY_THREADS)
k_lag_bary(const uint4* __restrict__ evals, uint32_t len, uint32_t base, NttTables tb, const int32_t* __restrict__ inv, int32_t* __restrict__ partial) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t i0 = blockIdx.x * LAG_BLOCK + threadIdx.x;
    Fr sum;
    fe_set_zero(sum);
#pragma unroll
    for (int k = 0; k < LAG_PER; ++k) {
        const uint32_t i = i0 + (uint32_t)k * POLY_THREADS;
        if (i < len) {
            Fr f, w, iv, term;
            wire_load(f, evals, i);
            pl_load(iv, inv, len, i);
            domain_elem(w, tb, base + i);
            fe_mul(term, f, w);
            fe_mul(term, term, iv);
            fe_sub(sum, sum, term);                          // f_i w^i / (z - w^i) = -(f_i w^i inv_i)
            fe_norm(sum);
        }
    }
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x == 0) pl_store(partial, gridDim.x, blockIdx.x, sum);
}

// ---- L2: sum of the per-workgroup partials -> one wire element -----------------------------------------------------------------------
__global__ void __launch_bounds__(POLY_THREADS)
k_lag_sum(const int32_t* __restrict__ partial, uint32_t n_partial, uint4* __restrict__ out_wire) {
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
    if (threadIdx.x == 0) wire_store(out_wire, 0, sum);
}

// ---- L3: quotient evaluations of the slice, q_i = (f_i - y) inv_i (kzg.rs:151-174); the on-domain element (slice index m_slice, if
// this slice owns it) is written as ZERO -- its term q_m L_m is added by the fold (kzg.rs:237-260 needs every rank's T = sum q_i w^i) ----
__global__ void __launch_bounds__(POLY_THREADS)
k_lag_quotient(const uint4* __restrict__ evals, uint32_t len, uint32_t base, NttTables tb, const int32_t* __restrict__ inv,
               const uint4* __restrict__ y_wire, uint32_t m_slice, int on_domain, uint4* __restrict__ q_out, int32_t* __restrict__ partial) {
    __shared__ int32_t lds[NL * POLY_THREADS];
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr y;
    wire_load(y, y_wire, 0);
    Fr sum;
    fe_set_zero(sum);
    uint32_t cnt = 0;
    for (uint32_t i = t; i < len; i += T, ++cnt) {
        if (i == m_slice) {
            q_out[2 * (size_t)i] = make_uint4(0, 0, 0, 0);
            q_out[2 * (size_t)i + 1] = make_uint4(0, 0, 0, 0);
            continue;
        }
        Fr f, iv, q;
        wire_load(f, evals, i);
        pl_load(iv, inv, len, i);
        fe_sub(f, f, y);                                     // (-3m, 3m)
        fe_mul(q, f, iv);
        wire_store(q_out, i, q);
        if (on_domain) {
            Fr w, term;
            domain_elem(w, tb, base + i);
            fe_mul(term, q, w);
            fe_add(sum, sum, term);
            fe_norm(sum);
            if (cnt % 32 == 31) fe_reduce(sum);
        }
    }
    if (!on_domain) return;                                  // uniform across the grid
    fe_reduce(sum);
    block_sum(sum, lds);
    if (threadIdx.x == 0) pl_store(partial, gridDim.x, blockIdx.x, sum);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------
namespace {
// pinned layout of a slot's PolySet for this path: [0,32) z | [32,64) y | [64,96) S readback | [96,128) f_m readback | [128,160) T readback
// device `small`:  [0,32) z | [32,64) y | [64,68) on-domain slice index | [128,160) S | [160,192) T | [4096, ..) per-workgroup partials
constexpr size_t LAG_SMALL_PARTIALS = 4096;
}  // namespace

// The context's AUXILIARY stream (high priority, created on first use; engine.h ctx_aux_stream): short latency-bound kernel sequences that other
// work waits for.  Phase 1 of every Lagrange-sharded proof of a context (upload, inverses, partial sum: ~0.1 ms on few waves) runs here, and so
// does the coset inversion chain of poly.hip's proofs (beside the upload of the evaluations).  On the slot's stream it shared a hardware queue with other slots' MSM kernels (HIP maps its streams onto a
// handful of queues) and sat behind them: in a stream of 2^17-element proofs the host waited 0.45 ms per blob for a 0.1 ms phase
// (tools/trace_config4_stream.py).
int32_t ctx_aux_stream(kzg_ctx* ctx, hipStream_t slot_stream, hipStream_t* out) {
    (void)slot_stream;
    if (!ctx->lag_stream) {
        int least = 0, greatest = 0;
        KZG_HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        KZG_HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->lag_stream, hipStreamNonBlocking, greatest));
    }
    *out = ctx->lag_stream;
    return KZG_OK;
}

// commit_slot >= 0: the slice's COMMITMENT (MSM of the evaluations themselves over the same shard, kzg.rs:96-100) is enqueued on that slot as
// well, reading the copy this call uploads (one H2D for both) -- collect it with msm_end(commit_slot); the proof slot must not be begun
// again before that
int32_t lag_begin(kzg_ctx* ctx, const kzg_srs* shard, size_t base, const void* evals, bool on_device, size_t len, size_t n,
                  const uint64_t z[4], int slot, int commit_slot) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || commit_slot >= KZG_NUM_SLOTS) return KZG_ERR_INVALID_ARG;
    LagProof& lp = ctx->lag[slot];
    if (lp.phase != 0 || ctx->slot_pending[slot]) { ctx->last_error = "the slot is in flight"; return KZG_ERR_INVALID_ARG; }
    // commit_slot == slot: GROUPED -- the commitment's MSM is not launched here but together with the proof's, as ONE batched launch over the shard's
    // per-bit tables (two scalar sets, one kernel sequence: msm_begin_batch) once the quotient exists.  Shard-sized MSMs are bound by dependent
    // latency; two of them per launch cost 0.33 ms where two launches cost 0.46 (2^17 pairs each).  Needs the tables and room for two sets.
    const bool grouped = commit_slot == slot;
    if (grouped && len && (!srs_bits(shard) || msm_batch_capacity(len) < 2)) {
        ctx->last_error = "grouped commitment + proof needs the shard's per-bit tables and room for two scalar sets in one launch (kzg_msm_batch_capacity)";
        return KZG_ERR_INVALID_ARG;
    }
    if (commit_slot >= 0 && !grouped && (ctx->lag[commit_slot].phase != 0 || ctx->slot_pending[commit_slot])) { ctx->last_error = "the commitment's slot is in flight"; return KZG_ERR_INVALID_ARG; }
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    const int log_n = ilog2_sz(n);
    PolySet& set = ctx->poly[slot];
    if (!set.pinned) KZG_HIP_TRY(ctx, hipHostMalloc(&set.pinned, 4096, hipHostMallocDefault));
    uint8_t* pin = static_cast<uint8_t*>(set.pinned);
    // z on the domain?  z^n == 1; then m with w^m = z (one bit per step, ~log^2 n / 2 host products)
    uint64_t zn[4];
    h_pow2k(z, log_n, zn);
    lp = LagProof();
    lp.shard = shard; lp.base = base; lp.len = len; lp.n = n;
    lp.d_evals = on_device ? evals : nullptr;
    lp.grouped = grouped;
    lp.on_domain = h_is_one(zn);
    if (lp.on_domain && !h_domain_index(z, log_n, &lp.m)) { ctx->last_error = "z^n = 1 but z is no power of the domain generator"; return KZG_ERR_ROOT_NOT_FOUND; }
    memcpy(pin, z, 32);
    memset(pin + 64, 0, 96);
    if (len == 0) { lp.phase = 1; return KZG_OK; }             // an empty slice contributes zero sums and the identity (no commitment MSM is enqueued either)
    NttTables tb;
    rc = ntt_get_tables(ctx, log_n, false, &tb);
    if (rc != KZG_OK) { lp = LagProof(); return rc; }
    const uint32_t blocks = (uint32_t)((len + LAG_BLOCK - 1) / LAG_BLOCK);
    hipStream_t s1 = nullptr;                                    // phase 1's stream (high priority)
    bool commit_started = false;                                 // a failure after the commitment's MSM was enqueued collects it, so that no slot stays pending without an owner
    auto fail = [&](hipError_t e, const char* where) {
        lp = LagProof();
        (void)hipStreamSynchronize(st);
        if (s1) (void)hipStreamSynchronize(s1);
        if (commit_started) { uint64_t sink[16]; (void)msm_end(ctx, commit_slot, nullptr, nullptr, sink); }
        return set_error(ctx, e, where);
    };
#define LAG_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e, #expr); } while (0)
    LAG_TRY(set.a.reserve(len * 32 + 32));
    LAG_TRY(set.b.reserve(len * NL * 4 + 64));
    LAG_TRY(set.c.reserve(len * 32 + 32));
    LAG_TRY(set.small.reserve(LAG_SMALL_PARTIALS + (size_t)blocks * NL * 4 + 64));
    uint8_t* small = set.small.as<uint8_t>();
    const uint32_t no_index = NO_INDEX;
    rc = ctx_aux_stream(ctx, st, &s1);
    if (rc != KZG_OK) { lp = LagProof(); return rc; }
    if (!ctx->lag_phase1[slot]) LAG_TRY(hipEventCreateWithFlags(&ctx->lag_phase1[slot], hipEventDisableTiming));
    memcpy(pin + 160, &no_index, 4);
    LAG_TRY(hipMemcpyAsync(small, pin, 32, hipMemcpyHostToDevice, s1));
    LAG_TRY(hipMemcpyAsync(small + 64, pin + 160, 4, hipMemcpyHostToDevice, s1));
    // Evaluations in HOST memory: the inverses need z only, so they are enqueued FIRST (auxiliary stream) and run while this thread sits in the pageable upload
    // (the slot's stream); the sum waits for both.  Per-rank proof of a 2^17-element slice, one call at a time: 0.63 -> 0.56 ms.
    const bool split = !on_device && s1 != st;
    if (split) {
        hipLaunchKernelGGL(k_lag_inverses, dim3(blocks), dim3(POLY_THREADS), 0, s1, static_cast<const uint4*>(nullptr), (uint32_t)len, (uint32_t)base, tb,
                           reinterpret_cast<const uint4*>(small), set.b.as<int32_t>(), reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS),
                           reinterpret_cast<uint32_t*>(small + 64));
        LAG_TRY(hipGetLastError());
        if (!ctx->lag_uploaded[slot]) LAG_TRY(hipEventCreateWithFlags(&ctx->lag_uploaded[slot], hipEventDisableTiming));
        LAG_TRY(hipMemcpyAsync(set.a.p, evals, len * 32, hipMemcpyHostToDevice, st));
        LAG_TRY(hipEventRecord(ctx->lag_uploaded[slot], st));
        LAG_TRY(hipStreamWaitEvent(s1, ctx->lag_uploaded[slot], 0));
    } else if (!on_device) {
        LAG_TRY(hipMemcpyAsync(set.a.p, evals, len * 32, hipMemcpyHostToDevice, s1));
    }
    const uint4* d_ev = on_device ? static_cast<const uint4*>(evals) : set.a.as<uint4>();     // resident evaluations are read in place
    if (commit_slot >= 0 && !grouped) {                         // the commitment of the same slice on its own slot, behind the upload
        hipStream_t st_c = nullptr;
        rc = msm_slot_stream(ctx, commit_slot, &st_c);
        if (rc != KZG_OK) { lp = LagProof(); (void)hipStreamSynchronize(s1); return rc; }
        if (split) {
            if (st_c != st) LAG_TRY(hipStreamWaitEvent(st_c, ctx->lag_uploaded[slot], 0));      // (recorded on the proof slot's stream, behind the upload)
        } else if (!on_device && st_c != s1) {
            if (!ctx->lag_uploaded[slot]) LAG_TRY(hipEventCreateWithFlags(&ctx->lag_uploaded[slot], hipEventDisableTiming));
            LAG_TRY(hipEventRecord(ctx->lag_uploaded[slot], s1));
            LAG_TRY(hipStreamWaitEvent(st_c, ctx->lag_uploaded[slot], 0));
        }
        rc = msm_begin(ctx, commit_slot, srs_bases(shard, 0, len, ctx->msm_c_override == 0), d_ev, len);
        if (rc != KZG_OK) { lp = LagProof(); (void)hipStreamSynchronize(s1); return rc; }
        commit_started = true;
    }
    if (split)
        hipLaunchKernelGGL(k_lag_bary, dim3(blocks), dim3(POLY_THREADS), 0, s1, d_ev, (uint32_t)len, (uint32_t)base, tb, set.b.as<int32_t>(),
                           reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS));
    else
        hipLaunchKernelGGL(k_lag_inverses, dim3(blocks), dim3(POLY_THREADS), 0, s1, d_ev, (uint32_t)len, (uint32_t)base, tb,
                           reinterpret_cast<const uint4*>(small), set.b.as<int32_t>(), reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS),
                           reinterpret_cast<uint32_t*>(small + 64));
    hipLaunchKernelGGL(k_lag_sum, dim3(1), dim3(POLY_THREADS), 0, s1, reinterpret_cast<const int32_t*>(small + LAG_SMALL_PARTIALS), blocks,
                       reinterpret_cast<uint4*>(small + 128));
    LAG_TRY(hipGetLastError());
    LAG_TRY(hipMemcpyAsync(pin + 64, small + 128, 32, hipMemcpyDeviceToHost, s1));
    if (lp.on_domain && lp.m >= base && lp.m - base < len)     // this slice owns f_m = y (helpers.rs:497-504)
        LAG_TRY(hipMemcpyAsync(pin + 96, reinterpret_cast<const uint8_t*>(d_ev) + (size_t)(lp.m - base) * 32, 32, hipMemcpyDeviceToHost, s1));
    LAG_TRY(hipEventRecord(ctx->lag_phase1[slot], s1));            // phase 2 (the slot's stream) and lag_partial_y wait for THIS proof's phase 1 only
    lp.phase = 1;
    return KZG_OK;
}

int32_t lag_partial_y(kzg_ctx* ctx, int slot, uint64_t out[8]) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->lag[slot].phase != 1) return KZG_ERR_INVALID_ARG;
    LagProof& lp = ctx->lag[slot];
    hipError_t e = lp.len ? hipEventSynchronize(ctx->lag_phase1[slot]) : hipSuccess;
    if (e != hipSuccess) { lp = LagProof(); return set_error(ctx, e, "lagrange proof: partial sum"); }
    const uint8_t* pin = static_cast<const uint8_t*>(ctx->poly[slot].pinned);
    memcpy(out, pin + 64, 64);                                // S_g | f_m (zero unless this slice owns m)
    if (lp.on_domain) memset(out, 0, 32);                     // the barycentric sum is not used for a domain point (and holds a dummy term)
    lp.phase = 2;
    return KZG_OK;
}

int32_t lag_continue(kzg_ctx* ctx, int slot, const uint64_t y[4]) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->lag[slot].phase != 2) return KZG_ERR_INVALID_ARG;
    LagProof& lp = ctx->lag[slot];
    if (lp.len == 0) { lp.phase = 3; return KZG_OK; }
    hipStream_t st = nullptr;
    (void)msm_slot_stream(ctx, slot, &st);
    PolySet& set = ctx->poly[slot];
    uint8_t* pin = static_cast<uint8_t*>(set.pinned);
    uint8_t* small = set.small.as<uint8_t>();
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, ilog2_sz(lp.n), false, &tb);
    if (rc != KZG_OK) { lp = LagProof(); return rc; }
    auto fail = [&](hipError_t e, const char* where) { lp = LagProof(); (void)hipStreamSynchronize(st); return set_error(ctx, e, where); };
    memcpy(pin + 32, y, 32);
    LAG_TRY(hipStreamWaitEvent(st, ctx->lag_phase1[slot], 0));     // the inverses (already complete: lag_partial_y waited for the event)
    LAG_TRY(hipMemcpyAsync(small + 32, pin + 32, 32, hipMemcpyHostToDevice, st));
    const bool owner = lp.on_domain && lp.m >= lp.base && lp.m - lp.base < lp.len;
    const uint32_t m_slice = owner ? (uint32_t)(lp.m - lp.base) : NO_INDEX;
    // same lanes-per-element shape as k_poly_quotient: four elements per lane
    uint32_t blocks = (uint32_t)((lp.len + LAG_BLOCK - 1) / LAG_BLOCK);
    const uint4* d_ev = lp.d_evals ? static_cast<const uint4*>(lp.d_evals) : set.a.as<uint4>();
    hipLaunchKernelGGL(k_lag_quotient, dim3(blocks), dim3(POLY_THREADS), 0, st, d_ev, (uint32_t)lp.len, (uint32_t)lp.base, tb,
                       set.b.as<int32_t>(), reinterpret_cast<const uint4*>(small + 32), m_slice, lp.on_domain ? 1 : 0, set.c.as<uint4>(),
                       reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS));
    if (lp.on_domain) {
        hipLaunchKernelGGL(k_lag_sum, dim3(1), dim3(POLY_THREADS), 0, st, reinterpret_cast<const int32_t*>(small + LAG_SMALL_PARTIALS), blocks,
                           reinterpret_cast<uint4*>(small + 160));
        LAG_TRY(hipMemcpyAsync(pin + 128, small + 160, 32, hipMemcpyDeviceToHost, st));
    }
    LAG_TRY(hipGetLastError());
#undef LAG_TRY
    if (lp.grouped) {                                           // commitment (the evaluations) and proof (the quotient) as ONE launch, in that order
        MsmBases b;
        b.points = srs_bits(lp.shard); b.table_stride = (uint32_t)lp.shard->n; b.c = 7; b.W = 255; b.naf = true;
        const void* sets[2] = {d_ev, set.c.p};
        rc = msm_begin_batch(ctx, slot, b, sets, lp.len, 2);
    } else {
        rc = msm_begin(ctx, slot, srs_bases(lp.shard, 0, lp.len, ctx->msm_c_override == 0), set.c.p, lp.len);
    }
    if (rc != KZG_OK) { lp = LagProof(); (void)hipStreamSynchronize(st); return rc; }
    lp.msm_started = true;
    lp.phase = 3;
    return KZG_OK;
}

// out_part (32 words): [0,16) XYZZ partial | [16,20) T = sum_{i in slice, i != m} q_i w^i (z on the domain, else 0) | [20,28) L_m (wire; the
// owner of m only, else 0) | [28] 1 if this slice owns m | [29,32) 0
// out_commit (grouped launches only, else ignored): the commitment's XYZZ partial, 16 words
int32_t lag_end(kzg_ctx* ctx, int slot, uint64_t out_part[32], uint64_t* out_commit) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->lag[slot].phase != 3) return KZG_ERR_INVALID_ARG;
    LagProof lp = ctx->lag[slot];
    if (lp.grouped && !out_commit) return KZG_ERR_INVALID_ARG;           // (left in flight: collect it with kzg_commit_and_prove_lagrange_end)
    ctx->lag[slot] = LagProof();
    memset(out_part, 0, 256);
    if (out_commit) memset(out_commit, 0, 128);
    if (lp.len == 0) return KZG_OK;
    int32_t rc;
    if (lp.grouped) {
        uint64_t both[32];
        rc = msm_end_batch(ctx, slot, 2, nullptr, nullptr, both);
        if (rc == KZG_OK) { memcpy(out_commit, both, 128); memcpy(out_part, both + 16, 128); }
    } else {
        rc = msm_end(ctx, slot, nullptr, nullptr, out_part);             // waits for the slot's stream: T is in the pinned buffer as well
    }
    if (rc != KZG_OK) return rc;
    if (!lp.on_domain) return KZG_OK;
    const uint8_t* pin = static_cast<const uint8_t*>(ctx->poly[slot].pinned);
    memcpy(out_part + 16, pin + 128, 32);
    if (lp.m >= lp.base && lp.m - lp.base < lp.len) {
        rc = srs_download(ctx, lp.shard->d_points + 4 * (lp.m - lp.base), 1, out_part + 20);
        if (rc != KZG_OK) return rc;
        out_part[28] = 1;
    }
    return KZG_OK;
}

// KZG::compute_quotient_eval_on_domain (prover/src/kzg.rs:237-260): sum over the n roots w^i != z of (f_i - value) w^i / ((z - w^i) z), the quotient's
// evaluation AT the domain point z -- here as -(1/z) T with T = sum_{w^i != z} q_i w^i, q_i = (f_i - value) / (w^i - z): the two kernels of a slice
// (base 0, len n) and the sum, on slot 0's working set; the reference does not require z to be a domain point (then no term is skipped).
// z = 0: the reference divides by zero (a panic in ark-ff) -> KZG_ERR_INVALID_ARG.
int32_t lag_quotient_eval_on_domain(kzg_ctx* ctx, const uint64_t z[4], const uint64_t* evals, size_t n, const uint64_t value[4], uint64_t out[4]) {
    const int slot = 0;
    if (ctx->lag[slot].phase != 0 || ctx->slot_pending[slot]) { ctx->last_error = "slot 0 is in flight"; return KZG_ERR_INVALID_ARG; }
    static const uint64_t zero[4] = {0, 0, 0, 0};
    if (memcmp(z, zero, 32) == 0) { ctx->last_error = "compute_quotient_eval_on_domain: z = 0 (the reference divides by z)"; return KZG_ERR_INVALID_ARG; }
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    const int log_n = ilog2_sz(n);
    NttTables tb;
    rc = ntt_get_tables(ctx, log_n, false, &tb);
    if (rc != KZG_OK) return rc;
    PolySet& set = ctx->poly[slot];
    if (!set.pinned) KZG_HIP_TRY(ctx, hipHostMalloc(&set.pinned, 4096, hipHostMallocDefault));
    uint8_t* pin = static_cast<uint8_t*>(set.pinned);
    const uint32_t blocks = (uint32_t)((n + LAG_BLOCK - 1) / LAG_BLOCK);
    auto fail = [&](hipError_t e, const char* where) { (void)hipStreamSynchronize(st); return set_error(ctx, e, where); };
#define LAG_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e, #expr); } while (0)
    LAG_TRY(set.a.reserve(n * 32 + 32));
    LAG_TRY(set.b.reserve(n * NL * 4 + 64));
    LAG_TRY(set.c.reserve(n * 32 + 32));
    LAG_TRY(set.small.reserve(LAG_SMALL_PARTIALS + (size_t)blocks * NL * 4 + 64));
    uint8_t* small = set.small.as<uint8_t>();
    memcpy(pin, z, 32);
    memcpy(pin + 32, value, 32);
    const uint32_t no_index = NO_INDEX;
    memcpy(pin + 160, &no_index, 4);
    LAG_TRY(hipMemcpyAsync(small, pin, 64, hipMemcpyHostToDevice, st));
    LAG_TRY(hipMemcpyAsync(small + 64, pin + 160, 4, hipMemcpyHostToDevice, st));
    LAG_TRY(hipMemcpyAsync(set.a.p, evals, n * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_lag_inverses, dim3(blocks), dim3(POLY_THREADS), 0, st, set.a.as<uint4>(), (uint32_t)n, 0u, tb, reinterpret_cast<const uint4*>(small),
                       set.b.as<int32_t>(), reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS), reinterpret_cast<uint32_t*>(small + 64));
    LAG_TRY(hipMemcpyAsync(pin + 160, small + 64, 4, hipMemcpyDeviceToHost, st));
    LAG_TRY(hipStreamSynchronize(st));                          // the index of the root equal to z, if any (found by the kernel)
    uint32_t m_slice = NO_INDEX;
    memcpy(&m_slice, pin + 160, 4);
    hipLaunchKernelGGL(k_lag_quotient, dim3(blocks), dim3(POLY_THREADS), 0, st, set.a.as<uint4>(), (uint32_t)n, 0u, tb, set.b.as<int32_t>(),
                       reinterpret_cast<const uint4*>(small + 32), m_slice, 1, set.c.as<uint4>(), reinterpret_cast<int32_t*>(small + LAG_SMALL_PARTIALS));
    hipLaunchKernelGGL(k_lag_sum, dim3(1), dim3(POLY_THREADS), 0, st, reinterpret_cast<const int32_t*>(small + LAG_SMALL_PARTIALS), blocks,
                       reinterpret_cast<uint4*>(small + 160));
    LAG_TRY(hipGetLastError());
    LAG_TRY(hipMemcpyAsync(pin + 128, small + 160, 32, hipMemcpyDeviceToHost, st));
    LAG_TRY(hipStreamSynchronize(st));
#undef LAG_TRY
    uint64_t t[4], zi[4], prod[4];
    memcpy(t, pin + 128, 32);
    h_fr_inv(z, zi);
    h_fr_mul(t, zi, prod);
    h_fr_sub(zero, prod, out);
    return KZG_OK;
}

// the slot gives up whatever it has in flight (error paths of the hosts above: a peer failed between two phases)
void lag_abort(kzg_ctx* ctx, int slot) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS) return;
    LagProof lp = ctx->lag[slot];
    ctx->lag[slot] = LagProof();
    hipStream_t st = nullptr;
    (void)msm_slot_stream(ctx, slot, &st);
    if (lp.phase >= 1 && lp.len && ctx->lag_phase1[slot]) (void)hipEventSynchronize(ctx->lag_phase1[slot]);
    if (lp.msm_started && ctx->slot_pending[slot]) {
        uint64_t sink[32];
        if (lp.grouped) (void)msm_end_batch(ctx, slot, 2, nullptr, nullptr, sink);
        else (void)msm_end(ctx, slot, nullptr, nullptr, sink);
    }
    else if (st) (void)hipStreamSynchronize(st);
}

}  // namespace kzg
