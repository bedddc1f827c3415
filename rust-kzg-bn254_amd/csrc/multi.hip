// multi.hip — the multi-GPU split of SURVEY.md §8e behind the C-ABI, without torch: one context, one resident SRS shard and one
// host thread per device; device g holds the SRS powers [g N / G, (g+1) N / G) and commits that slice of every polynomial; the G
// partial sums (128 B each) are folded on the host (G - 1 point additions + one inversion).  No inter-GPU traffic at all: the
// caller's buffers are read by every device's own H2D copy.  (bench.py --gpus N keeps the one-process-per-GPU / RCCL form that
// BASELINE.json's north_star prescribes: rust-kzg-bn254_amd/sharding.py.)
// Replaces, for a Rust host: the rayon-parallel `G1Projective::msm` of prover/src/kzg.rs:100,121 across several GPUs.
#include "engine.h"
#include "host_curve.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <cstdlib>
#include <string>
#include <new>
#include <thread>
#include <utility>
#include <vector>

constexpr int KZG_MULTI_BUFFERS = 16;    // resident scalar buffers per handle (kzg_multi_scalars_upload)
struct kzg_multi {
    std::vector<kzg_ctx*> ctx;
    std::vector<kzg_srs*> shard;
    std::vector<size_t> lo;          // first SRS power of shard g; lo[G] = N
    size_t n_total = 0;
    std::mutex mu;
    // resident scalars: buffer b of device g holds the coefficients [lo[g], min(lo[g+1], n_b)) of polynomial b
    std::vector<std::vector<void*>> dbuf;         // [g][b]
    size_t buf_n[KZG_MULTI_BUFFERS] = {};
    // evaluation-index shards of the Lagrange basis of lag_n points (kzg_multi_cache_lagrange): device g holds L_i, i in [elo[g], elo[g+1])
    std::vector<kzg_srs*> lag_shard;
    std::vector<size_t> elo;
    size_t lag_n = 0;
};

namespace {

template <class Fn>
int32_t for_each_device(kzg_multi* m, Fn fn) {
    const size_t G = m->ctx.size();
    std::vector<int32_t> rc(G, KZG_OK);
    std::vector<std::thread> th;
    th.reserve(G);
    for (size_t g = 1; g < G; ++g) th.emplace_back([&, g] { rc[g] = fn(g); });
    rc[0] = fn(0);
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; ++g) if (rc[g] != KZG_OK) return rc[g];
    return KZG_OK;
}
void set_bounds(kzg_multi* m, size_t n) {
    const size_t G = m->ctx.size();
    m->lo.assign(G + 1, 0);
    for (size_t g = 0; g <= G; ++g) m->lo[g] = g * n / G;
    m->n_total = n;
}
void drop_shards(kzg_multi* m) {
    for (auto& s : m->shard) { if (s) kzg_srs_free(s); s = nullptr; }
}
void drop_lagrange(kzg_multi* m) {
    for (auto& l : m->lag_shard) { if (l) kzg_srs_free(l); l = nullptr; }
    m->lag_n = 0;
}
void drop_buffers(kzg_multi* m) {
    for (size_t g = 0; g < m->dbuf.size(); ++g) {
        (void)hipSetDevice(m->ctx[g]->device);
        for (auto& p : m->dbuf[g]) { if (p) (void)hipFree(p); p = nullptr; }
    }
    for (auto& n : m->buf_n) n = 0;
}

}  // namespace

extern "C" {

int32_t kzg_multi_create(const int32_t* device_ids, int32_t n_devices, kzg_multi** out) {
    if (!out || !device_ids || n_devices <= 0 || n_devices > 64) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    kzg_multi* m = new (std::nothrow) kzg_multi();
    if (!m) return KZG_ERR_INVALID_ARG;
    for (int32_t i = 0; i < n_devices; ++i) {
        kzg_ctx* c = nullptr;
        int32_t rc = kzg_ctx_create(device_ids[i], &c);
        if (rc != KZG_OK) { for (auto* x : m->ctx) kzg_ctx_destroy(x); delete m; return rc; }
        m->ctx.push_back(c);
    }
    m->shard.assign((size_t)n_devices, nullptr);
    m->lag_shard.assign((size_t)n_devices, nullptr);
    m->dbuf.assign((size_t)n_devices, std::vector<void*>(KZG_MULTI_BUFFERS, nullptr));
    *out = m;
    return KZG_OK;
}

void kzg_multi_destroy(kzg_multi* m) {
    if (!m) return;
    drop_buffers(m);
    drop_lagrange(m);
    drop_shards(m);
    for (auto* c : m->ctx) kzg_ctx_destroy(c);
    delete m;
}

int32_t kzg_multi_device_count(const kzg_multi* m) { return m ? (int32_t)m->ctx.size() : 0; }
size_t kzg_multi_srs_len(const kzg_multi* m) { return m ? m->n_total : 0; }

int32_t kzg_multi_srs_upload(kzg_multi* m, const uint64_t* g1_xy_mont, size_t n_points) {
    if (!m || (n_points && !g1_xy_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    drop_lagrange(m);
    drop_shards(m);
    set_bounds(m, n_points);
    return for_each_device(m, [&](size_t g) {
        return kzg_srs_upload(m->ctx[g], g1_xy_mont + 8 * m->lo[g], m->lo[g + 1] - m->lo[g], &m->shard[g]);
    });
}

int32_t kzg_multi_srs_generate(kzg_multi* m, const uint64_t tau_mont[4], size_t n_points) {
    if (!m || !tau_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    drop_lagrange(m);
    drop_shards(m);
    set_bounds(m, n_points);
    return for_each_device(m, [&](size_t g) {
        return kzg_srs_generate(m->ctx[g], tau_mont, (uint64_t)m->lo[g], m->lo[g + 1] - m->lo[g], &m->shard[g]);
    });
}

static int32_t fold(const std::vector<uint64_t>& parts, size_t G, uint64_t out_xy[8], uint8_t* out_inf) {
    return kzg_g1_fold_partials(parts.data(), G, out_xy, out_inf);
}

int32_t kzg_multi_commit_coeff_form(kzg_multi* m, const uint64_t* coeffs_mont, size_t n, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!m || !out_xy_mont || (n && !coeffs_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n > m->n_total) return KZG_ERR_POLY_LENGTH;                                  // kzg.rs:112-116
    const size_t G = m->ctx.size();
    std::vector<uint64_t> parts(16 * G, 0);
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        const size_t lo = m->lo[g], hi = std::min(m->lo[g + 1], n);
        if (lo >= hi) return KZG_OK;                                                 // identity partial (zeros)
        return kzg_msm_g1_srs_partial(m->ctx[g], m->shard[g], 0, coeffs_mont + 4 * lo, hi - lo, parts.data() + 16 * g);
    });
    if (rc != KZG_OK) return rc;
    return fold(parts, G, out_xy_mont, out_is_infinity);
}

int32_t kzg_multi_commit_eval_form(kzg_multi* m, const uint64_t* evals_mont, size_t n, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!m || !out_xy_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n > m->n_total) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                        // kzg.rs:89-94
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    const size_t G = m->ctx.size();
    std::vector<uint64_t> parts(16 * G, 0);
    if (m->lag_n == n) {                                                             // sharded by evaluation index: every device reads ITS slice only, no IFFT
        int32_t rcl = for_each_device(m, [&](size_t g) -> int32_t {
            const size_t lo = m->elo[g], len = m->elo[g + 1] - lo;
            if (len == 0) return KZG_OK;
            return kzg_commit_eval_form_lagrange_partial(m->ctx[g], m->lag_shard[g], evals_mont + 4 * lo, len, parts.data() + 16 * g);
        });
        if (rcl != KZG_OK) return rcl;
        return fold(parts, G, out_xy_mont, out_is_infinity);
    }
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        if (m->lo[g] >= n || m->lo[g + 1] == m->lo[g]) return KZG_OK;
        return kzg_commit_eval_form_partial(m->ctx[g], m->shard[g], m->lo[g], evals_mont, n, parts.data() + 16 * g);
    });
    if (rc != KZG_OK) return rc;
    return fold(parts, G, out_xy_mont, out_is_infinity);
}

int32_t kzg_multi_compute_proof(kzg_multi* m, const uint64_t* evals_mont, size_t n, size_t n_roots, const uint64_t z_mont[4],
                                uint64_t out_xy_mont[8], uint8_t* out_is_infinity, uint64_t* out_y_mont) {
    if (!m || !out_xy_mont || !z_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;                                   // kzg.rs:135-139
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > m->n_total) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    const size_t G = m->ctx.size();
    if (m->lag_n == n) {                                                             // the four steps of kzg_compute_proof_lagrange_* with a host join in the middle
        std::vector<uint64_t> yparts(KZG_LAGRANGE_YPART_WORDS * G, 0), pparts(KZG_LAGRANGE_PART_WORDS * G, 0);
        std::vector<int32_t> started(G, 0);
        int32_t rcl = for_each_device(m, [&](size_t g) -> int32_t {
            const size_t lo = m->elo[g], len = m->elo[g + 1] - lo;
            int32_t r = kzg_compute_proof_lagrange_begin(m->ctx[g], m->lag_shard[g], lo, len ? evals_mont + 4 * lo : nullptr, len, n, z_mont, 0);
            if (r != KZG_OK) return r;
            started[g] = 1;
            r = kzg_compute_proof_lagrange_partial_y(m->ctx[g], 0, yparts.data() + KZG_LAGRANGE_YPART_WORDS * g);
            if (r != KZG_OK) started[g] = 0;
            return r;
        });
        uint64_t y[4] = {0, 0, 0, 0};
        if (rcl == KZG_OK) rcl = kzg_lagrange_fold_y(yparts.data(), G, n, z_mont, y);
        if (rcl != KZG_OK) { for (size_t g = 0; g < G; ++g) if (started[g]) (void)kzg_compute_proof_lagrange_abort(m->ctx[g], 0); return rcl; }
        rcl = for_each_device(m, [&](size_t g) -> int32_t {
            int32_t r = kzg_compute_proof_lagrange_continue(m->ctx[g], 0, y);
            if (r == KZG_OK) r = kzg_compute_proof_lagrange_end(m->ctx[g], 0, pparts.data() + KZG_LAGRANGE_PART_WORDS * g);
            if (r != KZG_OK) (void)kzg_compute_proof_lagrange_abort(m->ctx[g], 0);
            return r;
        });
        if (rcl != KZG_OK) return rcl;
        if (out_y_mont) memcpy(out_y_mont, y, 32);
        return kzg_lagrange_fold_proof(pparts.data(), G, n, z_mont, out_xy_mont, out_is_infinity);
    }
    std::vector<uint64_t> parts(16 * G, 0), ys(4 * G, 0);
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        if (g != 0 && (m->lo[g] >= n || m->lo[g + 1] == m->lo[g])) return KZG_OK;   // device 0 always runs: it reports y
        return kzg_compute_proof_partial(m->ctx[g], m->shard[g], m->lo[g], evals_mont, n, nullptr, n_roots, z_mont, parts.data() + 16 * g,
                                         ys.data() + 4 * g);
    });
    if (rc != KZG_OK) return rc;
    if (out_y_mont) memcpy(out_y_mont, ys.data(), 32);
    return fold(parts, G, out_xy_mont, out_is_infinity);
}

// Lagrange basis of the first n powers, sharded by EVALUATION index (device g keeps L_i for i in [g n / G, (g+1) n / G) with its own tables).
// One-time set-up: the powers are collected from the devices' shards, device 0 runs KZG::g1_ifft(n) (kzg.rs:263-285) over a plain copy,
// every device uploads its slice of the result.  From then on kzg_multi_commit_eval_form / kzg_multi_compute_proof of exactly n
// evaluations read only the device's own slice of the caller's buffer: no replicated upload, no IFFT, no whole-polynomial quotient.
int32_t kzg_multi_cache_lagrange(kzg_multi* m, size_t n) {
    if (!m) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;              // kzg.rs:265-269
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n > m->n_total) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    if (m->lag_n == n) return KZG_OK;
    drop_lagrange(m);
    const size_t G = m->ctx.size();
    std::vector<uint64_t> pts(n * 8);
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        const size_t lo = m->lo[g], hi = std::min(m->lo[g + 1], n);
        if (lo >= hi) return KZG_OK;
        return kzg_srs_download(m->ctx[g], m->shard[g], 0, hi - lo, pts.data() + 8 * lo);
    });
    if (rc != KZG_OK) return rc;
    kzg_srs* plain = nullptr;
    rc = kzg::srs_upload_plain(m->ctx[0], pts.data(), n, &plain);                    // no tables: g1_ifft of this size works on the points
    if (rc != KZG_OK) return rc;
    rc = kzg_g1_ifft(m->ctx[0], plain, n, pts.data());
    kzg_srs_free(plain);
    if (rc != KZG_OK) return rc;
    m->elo.assign(G + 1, 0);
    for (size_t g = 0; g <= G; ++g) m->elo[g] = g * n / G;
    rc = for_each_device(m, [&](size_t g) -> int32_t {
        return kzg_srs_upload(m->ctx[g], pts.data() + 8 * m->elo[g], m->elo[g + 1] - m->elo[g], &m->lag_shard[g]);
    });
    if (rc != KZG_OK) { drop_lagrange(m); return rc; }
    m->lag_n = n;
    return KZG_OK;
}

// ---- resident scalars and streams of commitments (what bench.py --multi times: inputs in HBM, several MSMs in flight per device) ----
int32_t kzg_multi_scalars_upload(kzg_multi* m, int32_t buffer_id, const uint64_t* coeffs_mont, size_t n) {
    if (!m || buffer_id < 0 || buffer_id >= KZG_MULTI_BUFFERS || (n && !coeffs_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n > m->n_total) return KZG_ERR_POLY_LENGTH;
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        kzg_ctx* c = m->ctx[g];
        KZG_HIP_TRY(c, hipSetDevice(c->device));
        void*& p = m->dbuf[g][(size_t)buffer_id];
        if (p) { (void)hipFree(p); p = nullptr; }
        const size_t lo = m->lo[g], hi = std::min(m->lo[g + 1], n);
        if (lo >= hi) return KZG_OK;
        KZG_HIP_TRY(c, hipMalloc(&p, (hi - lo) * 32));
        KZG_HIP_TRY(c, hipMemcpy(p, coeffs_mont + 4 * lo, (hi - lo) * 32, hipMemcpyHostToDevice));
        return KZG_OK;
    });
    if (rc == KZG_OK) m->buf_n[buffer_id] = n;
    return rc;
}

int32_t kzg_multi_commit_resident_stream(kzg_multi* m, const int32_t* buffer_ids, size_t count, uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    if (!m || (count && (!buffer_ids || !out_xy_mont))) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    for (size_t k = 0; k < count; ++k)
        if (buffer_ids[k] < 0 || buffer_ids[k] >= KZG_MULTI_BUFFERS || m->buf_n[buffer_ids[k]] == 0) return KZG_ERR_INVALID_ARG;
    const size_t G = m->ctx.size();
    std::vector<uint64_t> parts(16 * G * count, 0);                                   // [k][g]
    int32_t rc = for_each_device(m, [&](size_t g) -> int32_t {
        // this device's software pipeline: MSM k + depth - 1 is enqueued before MSM k is waited for (engine.h slots)
        const size_t shard_len = m->lo[g + 1] - m->lo[g];
        const int depth = shard_len >= ((size_t)1 << 20) ? 2 : 3;
        // (step, slot) pairs, oldest first.  Slots come from this device's own LAUNCH counter, not from the step index: steps whose
        // buffer does not reach this device's shard start nothing here, and k % depth would then hand a busy slot to a later step
        std::vector<std::pair<size_t, int32_t>> inflight;
        size_t launches = 0;
        auto retire = [&]() -> int32_t {
            const auto [k, slot] = inflight.front();
            inflight.erase(inflight.begin());
            return kzg_msm_g1_srs_end(m->ctx[g], slot, nullptr, nullptr, parts.data() + 16 * (k * G + g));
        };
        int32_t r = KZG_OK;
        for (size_t k = 0; k < count && r == KZG_OK; ++k) {
            const size_t n = m->buf_n[buffer_ids[k]], lo = m->lo[g], hi = std::min(m->lo[g + 1], n);
            if (lo >= hi) continue;                                                   // identity partial (zeros)
            if (inflight.size() == (size_t)depth) r = retire();                       // frees exactly the slot launch `launches` maps to
            if (r != KZG_OK) break;
            const int32_t slot = (int32_t)(launches % (size_t)depth);
            r = kzg_msm_g1_srs_device_begin(m->ctx[g], m->shard[g], 0, m->dbuf[g][(size_t)buffer_ids[k]], hi - lo, slot);
            if (r == KZG_OK) { inflight.emplace_back(k, slot); ++launches; }
        }
        while (!inflight.empty()) { const int32_t r2 = retire(); if (r == KZG_OK) r = r2; }   // drain whatever is in flight, also after an error
        return r;
    });
    if (rc != KZG_OK) return rc;
    for (size_t k = 0; k < count; ++k) {
        rc = kzg_g1_fold_partials(parts.data() + 16 * k * G, G, out_xy_mont + 8 * k, out_is_infinity ? out_is_infinity + k : nullptr);
        if (rc != KZG_OK) return rc;
    }
    return KZG_OK;
}

// ---- the exchanges over RCCL, behind the C-ABI (one process per GPU; north_star's "single RCCL all-reduce") ------------------------------
// RCCL has no reduction operator for elliptic-curve addition, so the all-reduce of G partial sums is an all-gather of G fixed-size rows
// and a fold on every rank.  The communicator belongs to the HOST (a Rust host creates it with its own RCCL binding: ncclGetUniqueId on
// rank 0, the id handed to the other ranks by whatever channel it has, ncclCommInitRank on this context's device); the library only
// issues the collective on its own stream.
// Which RCCL: the copy that is ALREADY in the process (the one the host created the communicator with) -- dlsym(RTLD_DEFAULT), then
// dlopen(.., RTLD_NOLOAD) of the usual names -- and only then a fresh load of /opt/rocm/lib/librccl.so (KZG_RCCL_LIB overrides every
// step).  No link-time dependency; single-GPU users never load it.  (a private RTLD_LOCAL copy need not be the instance
// the caller's ncclComm_t belongs to.)
//
// A row = 8 status bytes + payload.  FAILURE PROTOCOL: a rank whose local work failed still issues every collective of the call, with
// the status word POISON; it then returns its own error, every other rank returns KZG_ERR_PEER (last_error names the ranks) -- nobody
// is left blocked in ncclAllGather (round 4 returned before the collective).  A peer that never arrives: the wait on the stream is
// bounded by KZG_EXCHANGE_TIMEOUT_S (default 60 s) -> KZG_ERR_EXCHANGE_TIMEOUT; the communicator (and this context's stream) cannot
// be used afterwards, exit.
// The rows start in HOST memory: the partial sum of an MSM is produced by the host epilogue (bit-Horner over 208 points),
// and y / T of a proof are folded on the host; a row is <= 264 bytes, one pinned staging buffer each way.
}  // extern "C"
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
typedef int (*nccl_count_fn)(void*, int*);
struct Rccl {
    void* lib = nullptr; nccl_allgather_fn all_gather = nullptr; nccl_errstr_fn err = nullptr; nccl_count_fn count = nullptr, user_rank = nullptr;
    const char* how = "not loaded"; bool tried = false, have = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!r.tried) {
        r.tried = true;
        const char* env = kzg::opts().rccl_lib;
        if (env && *env) { r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL); r.have = r.lib != nullptr; r.how = "KZG_RCCL_LIB"; }
        if (!r.have && dlsym(RTLD_DEFAULT, "ncclAllGather")) { r.lib = RTLD_DEFAULT; r.have = true; r.how = "the RCCL already in the process (global scope)"; }
        if (!r.have) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) { r.have = true; r.how = "the RCCL already loaded by the host (RTLD_NOLOAD)"; break; } }
        }
        if (!r.have) {
            const char* names[] = {"/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
            for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) { r.have = true; r.how = "a fresh load of librccl.so"; break; } }
        }
        if (r.have) {
            r.all_gather = reinterpret_cast<nccl_allgather_fn>(dlsym(r.lib, "ncclAllGather"));
            r.err = reinterpret_cast<nccl_errstr_fn>(dlsym(r.lib, "ncclGetErrorString"));
            r.count = reinterpret_cast<nccl_count_fn>(dlsym(r.lib, "ncclCommCount"));
            r.user_rank = reinterpret_cast<nccl_count_fn>(dlsym(r.lib, "ncclCommUserRank"));
        }
    }
    return r;
}

constexpr uint64_t ROW_OK = 0, ROW_POISON = 0xFFFFFFFFFFFFFFFFULL;
constexpr size_t ROW_PAYLOAD_MAX = 256;

double exchange_timeout_s() {
    return kzg::opts().exchange_timeout_s;
}

// One all-gather of `bytes` <= 256 payload bytes per rank (+ the status word).  local_rc != KZG_OK: this rank sends POISON.
// gathered = world x bytes (payload only).  Returns KZG_OK, the rank's own local_rc (it was the one that failed), KZG_ERR_PEER,
// KZG_ERR_EXCHANGE_TIMEOUT, or an argument / device error.  Caller holds no lock; takes ctx->mu.
// *issued (optional): the collective went onto the stream -- the peers are (or will be) in it too.  A caller with a further collective in the same
// call must skip that one only when this one was NOT issued or timed out; a status that merely echoes local_rc says nothing about that.
int32_t rccl_exchange(kzg_ctx* ctx, void* comm, int32_t world, const void* payload, size_t bytes, int32_t local_rc, std::vector<uint64_t>& gathered,
                      bool* issued = nullptr) {
    if (issued) *issued = false;
    if (!ctx || !comm || world < 1 || world > 4096 || bytes > ROW_PAYLOAD_MAX || (bytes & 7)) return KZG_ERR_INVALID_ARG;
    Rccl& r = rccl();
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!r.all_gather) { ctx->last_error = "librccl.so could not be loaded (KZG_RCCL_LIB, the process's own copy, /opt/rocm/lib/librccl.so)"; return KZG_ERR_DEVICE; }
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (r.count) {                                                   // the staging buffer is sized by `world`: it must be the communicator's size
        int cnt = -1;
        if (r.count(comm, &cnt) != 0 || cnt != world) {
            char buf[160];
            snprintf(buf, sizeof buf, "world = %d but the communicator has %d rank(s)", (int)world, cnt);
            ctx->last_error = buf;
            return KZG_ERR_INVALID_ARG;
        }
    }
#ifdef KZG_TEST_HOOKS        // libkzg_bn254_mi355x_hooks.so only (make hooks; tests/test_gpu_rccl_cabi.py): a healthy rank that reports failure
    static const bool poison_self = []() { const char* e = getenv("KZG_RCCL_TEST_POISON"); return e && atoi(e) != 0; }();
#else
    constexpr bool poison_self = false;
#endif
    const size_t row = 8 + bytes, need = row * ((size_t)world + 1);
    kzg::DeviceBuffer& d = ctx->rccl_buf;
    KZG_HIP_TRY(ctx, d.reserve(need + 256));
    if (ctx->rccl_pinned_bytes < need) {
        if (ctx->rccl_pinned) { (void)hipHostFree(ctx->rccl_pinned); ctx->rccl_pinned = nullptr; ctx->rccl_pinned_bytes = 0; }
        KZG_HIP_TRY(ctx, hipHostMalloc(&ctx->rccl_pinned, need + 4096, hipHostMallocDefault));
        ctx->rccl_pinned_bytes = need + 4096;
    }
    char* pin = static_cast<char*>(ctx->rccl_pinned);
    const uint64_t status = (local_rc != KZG_OK || poison_self) ? ROW_POISON : ROW_OK;
    memcpy(pin, &status, 8);
    if (bytes) { if (local_rc == KZG_OK && payload) memcpy(pin + 8, payload, bytes); else memset(pin + 8, 0, bytes); }
    char* dev = d.as<char>();
    KZG_HIP_TRY(ctx, hipMemcpyAsync(dev, pin, row, hipMemcpyHostToDevice, ctx->stream));
    const int rc = r.all_gather(dev, dev + row, row, /* ncclUint8 */ 1, comm, ctx->stream);
    if (rc != 0) {
        (void)hipStreamSynchronize(ctx->stream);
        ctx->last_error = std::string("ncclAllGather: ") + (r.err ? r.err(rc) : "error");
        return KZG_ERR_DEVICE;
    }
    if (issued) *issued = true;
    KZG_HIP_TRY(ctx, hipMemcpyAsync(pin + row, dev + row, row * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
    // bounded wait: a peer that never issues its collective must not block this rank for ever
    {
        const auto t0 = std::chrono::steady_clock::now();
        const double limit = exchange_timeout_s();
        unsigned spins = 0;
        for (;;) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return kzg::set_error(ctx, q, "the exchange on the context's stream");
            (void)hipGetLastError();
            if (++spins > 4000) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                ctx->last_error = "the all-gather did not complete within KZG_EXCHANGE_TIMEOUT_S: a peer rank is gone or hung; the communicator is unusable";
                return KZG_ERR_EXCHANGE_TIMEOUT;
            }
        }
    }
    gathered.assign((size_t)world * (bytes / 8), 0);
    std::string bad;
    for (int32_t g = 0; g < world; ++g) {
        uint64_t st_g;
        memcpy(&st_g, pin + row * ((size_t)g + 1), 8);
        if (st_g != ROW_OK) { if (!bad.empty()) bad += ", "; bad += std::to_string(g); }
        if (bytes) memcpy(gathered.data() + (size_t)g * (bytes / 8), pin + row * ((size_t)g + 1) + 8, bytes);
    }
    if (local_rc != KZG_OK) return local_rc;                          // this rank's own failure (the others see KZG_ERR_PEER)
    if (!bad.empty()) { ctx->last_error = "rank(s) " + bad + " of the communicator failed in this call"; return KZG_ERR_PEER; }
    return KZG_OK;
}
}  // namespace
extern "C" {

int32_t kzg_rccl_allgather_fold(kzg_ctx* ctx, void* nccl_comm, int32_t world, const uint64_t partial_xyzz_mont[16],
                                uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !nccl_comm || world < 1 || world > 4096 || !partial_xyzz_mont || !out_xy_mont) return KZG_ERR_INVALID_ARG;
    std::vector<uint64_t> gathered;
    const int32_t rc = rccl_exchange(ctx, nccl_comm, world, partial_xyzz_mont, 128, KZG_OK, gathered);
    if (rc != KZG_OK) return rc;
    return kzg_g1_fold_partials(gathered.data(), (size_t)world, out_xy_mont, out_is_infinity);
}

int32_t kzg_commit_coeff_form_rccl(kzg_ctx* ctx, const kzg_srs* srs_shard, const void* d_coeffs_shard_mont, size_t n_shard, void* nccl_comm,
                                   int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !srs_shard || !nccl_comm || !out_xy_mont || world < 1 || world > 4096) return KZG_ERR_INVALID_ARG;
    uint64_t part[16] = {0};                                         // an empty shard contributes the identity (ZZ = 0)
    int32_t local = KZG_OK;
    if (n_shard) local = d_coeffs_shard_mont ? kzg_msm_g1_srs_partial_device(ctx, srs_shard, 0, d_coeffs_shard_mont, n_shard, part) : KZG_ERR_INVALID_ARG;
    std::vector<uint64_t> gathered;
    const int32_t rc = rccl_exchange(ctx, nccl_comm, world, part, 128, local, gathered);    // a failed rank still joins the collective
    if (rc != KZG_OK) return rc;
    return kzg_g1_fold_partials(gathered.data(), (size_t)world, out_xy_mont, out_is_infinity);
}

// KZG::commit_eval_form (prover/src/kzg.rs:84-104) of this rank's slice of the evaluations over its shard of the Lagrange basis + the
// exchange + the fold, in one call (BASELINE config 4's commitment)
static int32_t commit_eval_form_rccl_common(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const void* evals, bool on_device, size_t len, void* nccl_comm,
                                            int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !lagrange_shard || !nccl_comm || !out_xy_mont || world < 1 || world > 4096) return KZG_ERR_INVALID_ARG;
    uint64_t part[16] = {0};
    int32_t local = KZG_OK;
    if (len && !evals) local = KZG_ERR_INVALID_ARG;
    else if (len) local = on_device ? kzg_commit_eval_form_lagrange_partial_device(ctx, lagrange_shard, evals, len, part)
                                    : kzg_commit_eval_form_lagrange_partial(ctx, lagrange_shard, static_cast<const uint64_t*>(evals), len, part);
    std::vector<uint64_t> gathered;
    const int32_t rc = rccl_exchange(ctx, nccl_comm, world, part, 128, local, gathered);
    if (rc != KZG_OK) return rc;
    return kzg_g1_fold_partials(gathered.data(), (size_t)world, out_xy_mont, out_is_infinity);
}
int32_t kzg_commit_eval_form_rccl(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const uint64_t* evals_slice_mont, size_t len, void* nccl_comm,
                                  int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    return commit_eval_form_rccl_common(ctx, lagrange_shard, evals_slice_mont, false, len, nccl_comm, world, out_xy_mont, out_is_infinity);
}
int32_t kzg_commit_eval_form_rccl_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const void* d_evals_slice_mont, size_t len, void* nccl_comm,
                                         int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    return commit_eval_form_rccl_common(ctx, lagrange_shard, d_evals_slice_mont, true, len, nccl_comm, world, out_xy_mont, out_is_infinity);
}

// KZG::compute_proof_impl (kzg.rs:128-178, :237-260) of this rank's slice, both exchanges inside: S_g -> y, then the partial points
static int32_t compute_proof_rccl_common(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* evals, bool on_device, size_t len,
                                         size_t n, const uint64_t z_mont[4], void* nccl_comm, int32_t world, uint64_t out_xy_mont[8],
                                         uint8_t* out_is_infinity, uint64_t* out_y_mont) {
    if (!ctx || !lagrange_shard || !nccl_comm || !out_xy_mont || !z_mont || world < 1 || world > 4096) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;           // the same on every rank: nobody issues a collective
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    const int32_t slot = 0;
    uint64_t ypart[KZG_LAGRANGE_YPART_WORDS] = {0}, part[KZG_LAGRANGE_PART_WORDS] = {0}, y[4] = {0, 0, 0, 0};
    int32_t local = on_device ? kzg_compute_proof_lagrange_begin_device(ctx, lagrange_shard, shard_lo, evals, len, n, z_mont, slot)
                              : kzg_compute_proof_lagrange_begin(ctx, lagrange_shard, shard_lo, static_cast<const uint64_t*>(evals), len, n, z_mont, slot);
    bool in_flight = local == KZG_OK;
    if (local == KZG_OK) { local = kzg_compute_proof_lagrange_partial_y(ctx, slot, ypart); if (local != KZG_OK) in_flight = false; }
    std::vector<uint64_t> got;
    bool issued = false;
    const bool own_failure = local != KZG_OK;
    std::string own_error = own_failure ? ctx->last_error : std::string();
    int32_t rc = rccl_exchange(ctx, nccl_comm, world, ypart, sizeof ypart, local, got, &issued);
    // The second collective is skipped ONLY when the first never went out (bad communicator, no RCCL: the same on every rank) or timed out (the
    // communicator is gone).  A status that echoes this rank's own failure -- KZG_ERR_DEVICE from an allocation, KZG_ERR_INVALID_ARG from a slot
    // in flight -- came back AFTER the collective: the peers hold KZG_ERR_PEER and are about to issue the second one, so this rank joins it too.
    if (!issued || rc == KZG_ERR_EXCHANGE_TIMEOUT) { if (in_flight) (void)kzg_compute_proof_lagrange_abort(ctx, slot); return rc; }
    int32_t first = rc;                                               // KZG_OK, this rank's own error, KZG_ERR_PEER, or a device error behind the collective
    if (first == KZG_OK) {
        local = kzg_lagrange_fold_y(got.data(), (size_t)world, n, z_mont, y);
        if (local == KZG_OK) local = kzg_compute_proof_lagrange_continue(ctx, slot, y);
        if (local == KZG_OK) local = kzg_compute_proof_lagrange_end(ctx, slot, part);
        if (local != KZG_OK) (void)kzg_compute_proof_lagrange_abort(ctx, slot);
    } else {
        if (in_flight) (void)kzg_compute_proof_lagrange_abort(ctx, slot);
        local = first;
    }
    rc = rccl_exchange(ctx, nccl_comm, world, part, sizeof part, local == KZG_ERR_PEER ? KZG_OK : local, got, &issued);
    if (own_failure) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->last_error = own_error + (issued ? " [this rank joined both collectives of the call with poisoned rows]" : " [the second collective could not be issued]");
    }
    if (first != KZG_OK) return first;
    if (rc != KZG_OK) return rc;
    if (out_y_mont) memcpy(out_y_mont, y, 32);
    return kzg_lagrange_fold_proof(got.data(), (size_t)world, n, z_mont, out_xy_mont, out_is_infinity);
}
int32_t kzg_compute_proof_rccl(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont, size_t len, size_t n,
                               const uint64_t z_mont[4], void* nccl_comm, int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity,
                               uint64_t* out_y_mont) {
    return compute_proof_rccl_common(ctx, lagrange_shard, shard_lo, evals_slice_mont, false, len, n, z_mont, nccl_comm, world, out_xy_mont,
                                     out_is_infinity, out_y_mont);
}
int32_t kzg_compute_proof_rccl_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont, size_t len, size_t n,
                                      const uint64_t z_mont[4], void* nccl_comm, int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity,
                                      uint64_t* out_y_mont) {
    return compute_proof_rccl_common(ctx, lagrange_shard, shard_lo, d_evals_slice_mont, true, len, n, z_mont, nccl_comm, world, out_xy_mont,
                                     out_is_infinity, out_y_mont);
}

}  // extern "C"
