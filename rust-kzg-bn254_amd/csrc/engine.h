// engine.h — host-side objects behind the C-ABI (include/kzg_bn254_mi355x.h): one context per GPU,
// device-resident SRS, reusable workspaces.  No torch types, no CPU fallback.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include "../../include/kzg_bn254_mi355x.h"

namespace kzg {

struct DeviceBuffer {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        size_t cap = need + need / 8;
        hipError_t e = hipMalloc(&p, cap);
        if (e == hipSuccess) bytes = cap;
        return e;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct MsmWorkspace {
    DeviceBuffer scalars;      // staging for host scalars (n x 32 B)
    DeviceBuffer blob;         // staging for blob bytes (asynchronous commit_blob)
    DeviceBuffer bases;        // staging for ad-hoc bases (n x 64 B, device format)
    DeviceBuffer bases_wire;   // staging for ad-hoc bases in wire format
    DeviceBuffer digits, sorted, count, blockbase, sort_tmp, sort_key, sort_small, offs, block_sums, bucket, chunkS, chunkTmp, chunkA, out_wire;
    DeviceBuffer head, cont;   // accumulate partials: head[g] per bucket, cont[t] per lane (36 limb planes each; msm_kernels.h section 4)
    const void* count_zero_ptr = nullptr;   // `count` at this address holds zeros in its first count_zero_g words once the stream gets here (small table-mode sort: the scan clears what it read)
    uint32_t count_zero_g = 0;
    void* pinned_out = nullptr;   // pinned host buffer for window sums
    void* pinned_out_dev = nullptr;   // its device address: the reduction kernels store their results there directly
    // optional per-phase timing with HIP events on the launch stream (kzg_ctx_set_profiling)
    static constexpr int N_PHASES = 8;   // digits, sort (histograms + scan), scatter, (unused), accumulate, bucket sums + reduce level 1, reduce level 2, whole launch
    hipEvent_t ev[N_PHASES] = {};
    bool ev_ready = false;
    hipEvent_t ev_done = nullptr;   // recorded behind the result copy of the MSM in flight on this workspace
    hipEvent_t ev_sorted = nullptr; // recorded behind the sort of that MSM (in front of its accumulate kernel): the next MSM of the context starts behind it
    double phase_ms[N_PHASES] = {};
    uint64_t profiled_launches = 0;
    uint64_t profiled_pairs = 0;
    uint64_t profiled_entries = 0;   // sorted entries (= mixed additions) of the profiled launches
    void release();
};

// scratch of the polynomial kernels: a = evaluations / bytes->Fr output, b = inverses | denominators (limb planes),
// c = quotient / blob bytes, small = scalars and per-block partial sums; pinned = 4 KiB host staging (init image, z, y readback)
struct PolySet {
    DeviceBuffer a, b, c, small;
    void* pinned = nullptr;
    hipEvent_t ev_chain = nullptr;   // behind the inversion chain of a proof when it runs on the auxiliary stream (poly.hip proof_enqueue)
    void release() {
        a.release(); b.release(); c.release(); small.release();
        if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; }
        if (ev_chain) { (void)hipEventDestroy(ev_chain); ev_chain = nullptr; }
    }
};

struct NttWorkspace {
    DeviceBuffer data, tmp;
    void release() { data.release(); tmp.release(); }
};

}  // namespace kzg

namespace kzg {
struct MsmPending;
struct BlobStream;   // blob -> commitment + proof jobs in flight (blobstream.hip)
// a proof over a slice of the evaluations and the matching slice of a Lagrange basis, in flight on one slot (lagrange.hip)
struct LagProof {
    int phase = 0;                 // 0 idle; 1 inverses + partial barycentric sum enqueued; 2 partial collected, waiting for y; 3 quotient (+ MSM) enqueued
    const kzg_srs* shard = nullptr;
    const void* d_evals = nullptr; // the caller's resident evaluations (read in place), or nullptr: the slot's copy
    size_t base = 0, len = 0, n = 0;   // the slice holds the evaluations [base, base + len) of the n-point domain
    bool on_domain = false;        // z = w^m
    uint32_t m = 0;
    bool msm_started = false;      // len > 0: an MSM is pending on the slot
    bool grouped = false;          // commitment and proof leave as ONE batched launch (two scalar sets) on this slot
};
}
#ifndef KZG_NUM_SLOTS
#define KZG_NUM_SLOTS 4      // slots of the asynchronous calls (include/kzg_bn254_mi355x.h: KZG_NUM_SLOTS)
#endif

struct kzg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string last_error;
    int msm_c_override = 0;
    int msm_seg_override = 0;
    int reduction_lanes = 0;               // kzg_ctx_set_reduction_lanes: 0 automatic, 2 lane pairs, 4 lane quads
    bool profiling = false;
    bool lds_attr_set = false;
    bool poly_lds_attr_set = false;
    uint32_t acc_wave_slots = 3 * 1024;   // resident waves of the accumulate kernel on this device: 3 per SIMD x 4 SIMDs x CUs (set at kzg_ctx_create; make_plan takes 2 of the 3 per SIMD where that pays)
    kzg::MsmWorkspace msm;
    kzg::MsmWorkspace msm_x[KZG_NUM_SLOTS - 1];   // workspaces of slots 1.. of the asynchronous calls
    hipStream_t stream_x[KZG_NUM_SLOTS - 1] = {}; // one stream per slot (slot 0: `stream`), created on first use
    kzg::MsmPending* slot_pending[KZG_NUM_SLOTS] = {};   // what each slot of the asynchronous calls has in flight
    kzg::NttWorkspace ntt;
    int32_t* ondomain_inv[13] = {};      // [log n] -> 1 / (w^k - 1), k < n <= 4096 (limb planes): proofs at a known domain point (poly.hip)
    kzg::NttWorkspace ntt_x[KZG_NUM_SLOTS - 1];   // slots 1.. of the asynchronous commitment / proof calls
    void* vb_pinned = nullptr;           // pinned staging of the packed blobs of one batch verification (capi.hip), grown on demand
    size_t vb_pinned_bytes = 0;
    kzg::DeviceBuffer rccl_buf;          // this rank's partial + the gathered partials of kzg_rccl_allgather_fold (multi.hip)
    void* rccl_pinned = nullptr;         // pinned host staging of the same rows (row out | world rows in)
    size_t rccl_pinned_bytes = 0;
    kzg::PolySet poly[KZG_NUM_SLOTS];   // polynomial / proof pipeline scratch, one set per slot ([0] also serves the synchronous calls)
    kzg::LagProof lag[KZG_NUM_SLOTS];   // Lagrange-sharded proofs in flight (kzg_compute_proof_lagrange_*)
    hipStream_t lag_stream = nullptr;       // high-priority stream of phase 1 (upload, inverses, partial sum) of every Lagrange-sharded proof of this context
    hipEvent_t lag_phase1[KZG_NUM_SLOTS] = {};     // behind phase 1 of the proof on that slot
    hipEvent_t lag_uploaded[KZG_NUM_SLOTS] = {};   // behind the slice's upload on the proof slot's stream: the commitment on another slot starts after it
    kzg::MsmWorkspace& slot_msm(int slot) { return slot ? msm_x[slot - 1] : msm; }
    kzg::BlobStream* blob_stream = nullptr;    // kzg_commit_and_prove_blob_begin / _end: created on first use
    hipEvent_t last_sorted = nullptr;          // ev_sorted of the most recently enqueued MSM launch of this context ...
    hipStream_t last_sorted_stream = nullptr;  // ... and the stream it went to (msm.hip msm_enqueue)
    kzg::NttWorkspace& slot_ntt(int slot) { return slot ? ntt_x[slot - 1] : ntt; }
};

struct kzg_srs {
    kzg_ctx* ctx = nullptr;
    // An SRS may be used by several host threads and by every context of its GPU at once (the reference shares one SRS across threads:
    // prover/tests/kzg_test.rs:9-17).  Everything below that is built AFTER the handle was returned -- per-bit tables on the first batched
    // call, the x3 tables of g1_ifft, cached Lagrange bases -- is built and published under lazy_mu, fully synchronised before the pointer
    // is stored, and read through the accessors below (a reader sees "absent" or "complete", never a table under construction).
    mutable std::mutex lazy_mu;
    // device affine format (curve.h), 64 B per point.  Without precomputation: n points.  With precomputed
    // window tables (pre_W > 0): pre_W x n points, table w holds T_w[i] = 2^(pre_c * w) * P_i (table 0 = the SRS).
    uint4* d_points = nullptr;
    size_t n = 0;
    int pre_c = 0;
    int pre_W = 0;
    // second table set with narrower windows (small_c < pre_c) for MSMs of at most SRS_SMALL_MAX pairs (srs.hip); may be absent
    uint4* d_small = nullptr;
    int small_c = 0;
    int small_W = 0;
    // per-bit tables Bit_j[i] = 2^j P_i, j < 255, n points apart (srs.hip srs_build_bit_tables): the NAF mode of MSMs of >= SRS_NAF_MIN pairs; may be absent
    uint4* d_bits = nullptr;
    // 3 Bit_p[j] for p < 255, j < 256 (g1fft.hip: the x3 tables of the 64..256-point g1_ifft), built on first use; a cache, hence mutable
    mutable uint4* d_t3 = nullptr;
    mutable uint32_t t3_n = 0;           // points covered by d_t3: min(n, 2048)
    // Lagrange-basis copies of the first m points (KZG::g1_ifft(m), kzg.rs:263-285), built by kzg_srs_cache_lagrange and used by
    // the eval-form commitments of exactly m evaluations instead of IFFT + MSM over the monomial basis; owned by this SRS
    std::map<size_t, kzg_srs*> lagrange;
    size_t lagrange_of = 0;      // this handle IS a Lagrange basis of that many points (0: monomial SRS)
};

namespace kzg {

// Bases of one MSM: `points` = first base of the slice; table_stride > 0 selects the precomputed-table mode
// (tables `table_stride` points apart, window bits c, W tables).
struct MsmBases {
    const uint4* points = nullptr;
    uint32_t table_stride = 0;
    int c = 0;
    int W = 0;
    bool bitsum = false;  // tiny MSM: `points` = the per-bit tables, summed directly (k_bitsum_level1 / 2, msm_kernels.h section 6e)
    bool naf = false;     // `points` = the per-bit tables (Bit_j[i] = 2^j P_i, j < 255, table_stride points apart): width-(c + 1) NAF digits
};
constexpr int SRS_SMALL_C = 15;                       // window bits of the second table set
constexpr size_t SRS_SMALL_MAX = (size_t)1 << 13;     // MSMs of up to this many pairs use it
constexpr size_t SRS_NAF_MIN = (size_t)1 << 11;       // an SRS of at least this many points gets per-bit tables at upload (16 KiB per point; from 2^15 points for the NAF mode of large MSMs, below for the bit sums of tiny ones)
constexpr size_t MSM_NAF_MIN = (size_t)1 << 14;       // MSMs of at least this many pairs use them (width-w NAF digits): one at a time 2^14 0.296 -> 0.273 ms; 2^13 0.241 -> 0.273: not below
// bucket bits (c: 2^(c-1) buckets, NAF width c + 1) of an MSM of n pairs over the per-bit tables: the window policy of srs_precompute, by MSM length
inline int srs_naf_c(size_t n) {
    if (n < ((size_t)1 << 15)) return 13;                                       // (13: 32 digit words per scalar; 15..17: 16)
    // measured (tools/time_shard_inflight.py, c = 15 / 16 / 17, two or three MSMs in flight, ms per MSM): 2^17 0.257 / 0.264 / 0.284,
    // 2^18 0.362 / 0.355 / 0.404, 2^19 0.678 / 0.631 / 0.629, 2^20 1.307 / 1.227 / 1.188 (profiles/r03_naf.md)
    return n >= ((size_t)1 << 20) ? 17 : n >= ((size_t)1 << 18) ? 16 : 15;
}
// bases of an MSM of n pairs over srs[offset .. offset + n)
inline uint4* srs_bits(const kzg_srs* srs) { std::lock_guard<std::mutex> lk(srs->lazy_mu); return srs->d_bits; }
inline uint4* srs_t3(const kzg_srs* srs) { std::lock_guard<std::mutex> lk(srs->lazy_mu); return srs->d_t3; }
inline kzg_srs* srs_cached_lagrange(const kzg_srs* srs, size_t n) {
    std::lock_guard<std::mutex> lk(srs->lazy_mu);
    auto it = srs->lagrange.find(n);
    return it == srs->lagrange.end() ? nullptr : it->second;
}
inline MsmBases srs_bases(const kzg_srs* srs, size_t offset, size_t n, bool allow_tables) {
    MsmBases b;
    std::lock_guard<std::mutex> lk(srs->lazy_mu);
    b.points = srs->d_points + 4 * offset;
    if (allow_tables && srs->pre_W > 0) {
        b.table_stride = (uint32_t)srs->n; b.c = srs->pre_c; b.W = srs->pre_W;
        if (srs->d_small && n <= SRS_SMALL_MAX) { b.points = srs->d_small + 4 * offset; b.c = srs->small_c; b.W = srs->small_W; }
        // bit sums up to 4 096 pairs; measured, one commitment at a time: 2^9 0.108 -> 0.066 ms, 2^10 0.120 -> 0.077, 2^11 0.157 -> 0.100, 2^12 0.176 -> 0.135, 2^13 0.197 -> 0.204
        if (srs->d_bits && n <= 4096) {
            b.points = srs->d_bits + 4 * offset; b.c = 0; b.W = 255; b.bitsum = true;
        }
        if (srs->d_bits && n >= MSM_NAF_MIN) {
            b.points = srs->d_bits + 4 * offset; b.c = srs_naf_c(n); b.W = 255; b.naf = true;
        }
    }
    return b;
}

// MSM over device-resident points (device format) and device-resident scalars (wire format).
// Writes the affine result (or the XYZZ partial if out_xyzz != nullptr).
int32_t msm_run(kzg_ctx* ctx, const MsmBases& bases, const void* d_scalars, size_t n,
                uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz);

int32_t msm_slot_stream(kzg_ctx* ctx, int slot, hipStream_t* out);
int32_t msm_begin(kzg_ctx* ctx, int slot, const MsmBases& bases, const void* d_scalars, size_t n);
int32_t msm_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz);
void msm_drop_slots(kzg_ctx* ctx);
int32_t msm_run_batch(kzg_ctx* ctx, const uint4* d_points, const void* d_scalars, size_t n, uint32_t batch,
                      uint64_t* out_xy, uint8_t* out_inf);

// Precompute the window tables of an SRS in place (reallocates srs->d_points); no-op for small / huge SRS.
int32_t srs_precompute(kzg_ctx* ctx, kzg_srs* srs);
int32_t srs_build_bit_tables(kzg_ctx* ctx, kzg_srs* srs, bool force);
// `polys` commitments over the first n points of one SRS in one kernel sequence (bases: the SRS's per-bit tables; msm.hip)
size_t msm_batch_capacity(size_t n);
int32_t msm_begin_batch(kzg_ctx* ctx, int slot, const MsmBases& bases, const void* const* d_scalars, size_t n, size_t count);
int32_t msm_end_batch(kzg_ctx* ctx, int slot, size_t count, uint64_t* out_xy, uint8_t* out_inf, uint64_t* out_xyzz);
int32_t msm_run_batch_tables(kzg_ctx* ctx, const MsmBases& bases, const void* d_scalars, size_t n, size_t polys, uint64_t* out_xy, uint8_t* out_inf);

// wire affine points (device memory) -> device affine format (curve.h), asynchronous on ctx->stream
// d_off_curve_flag != nullptr: also check y^2 == x^3 + 3 of every non-identity point, *flag |= 1 on a violation (device word, zeroed by the caller)
int32_t points_wire_to_device(kzg_ctx* ctx, const uint4* d_wire, uint4* d_out, size_t n, uint32_t* d_off_curve_flag = nullptr);

// Two-level twiddle table of one domain: w^t (t < lo_len) and w^(t * lo_len) (t < hi_len), 9 limb planes each,
// internal Montgomery form.  Cached per (device, log n, direction).
struct NttTables {
    int32_t* lo = nullptr;
    int32_t* hi = nullptr;
    uint32_t lo_len = 0, hi_len = 0;
    int lo_bits = 0;
};
int32_t ntt_get_tables(kzg_ctx* ctx, int log_n, bool inverse, NttTables* out);

// In-place NTT on device data (wire format), natural order in/out.
int32_t ntt_run(kzg_ctx* ctx, void* d_data, size_t n, bool inverse, hipStream_t st = nullptr, NttWorkspace* ws = nullptr);

// synthetic SRS P_i = tau^i * G1 written to d_points (device format); device SRS -> wire on the host
int32_t srs_generate(kzg_ctx* ctx, const uint64_t tau_mont[4], uint64_t first_power, size_t n, uint4* d_points);
int32_t srs_download(kzg_ctx* ctx, const uint4* d_points, size_t n, uint64_t* out_xy);
int32_t srs_decompress(kzg_ctx* ctx, const uint8_t* bytes, size_t n, uint4* d_points, uint32_t* err_kind, uint32_t* err_index, bool ark_le = false);

// KZG::g1_ifft: Lagrange-basis SRS of size n (n a power of two <= srs->n), affine wire points to the host / left on the device
int32_t g1_ifft_run(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint64_t* out_xy);
int32_t g1_ifft_device(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint4* d_out, bool wire);

int32_t set_error(kzg_ctx* ctx, hipError_t e, const char* where);
// the context's high-priority auxiliary stream (lagrange.hip)
int32_t ctx_aux_stream(kzg_ctx* ctx, hipStream_t fallback, hipStream_t* out);
// the points only, no window / per-bit tables (set-up paths that need the points once: kzg_multi_cache_lagrange)
int32_t srs_upload_plain(kzg_ctx* ctx, const uint64_t* g1_xy_mont, size_t n_points, kzg_srs** out);

// job(i) for i < n on the library's persistent host pool (capi.hip HostPool; the calling thread takes part)
void host_parallel_for(size_t n, const std::function<void(size_t)>& job);
// joins the transcript threads and frees the buffers of the blob stream (kzg_ctx_destroy)
void blob_stream_release(kzg_ctx* ctx);

// Every environment variable the library reads (capi.hip opts(); header: "ENVIRONMENT").  Read once per process, except the two SRS switches,
// which are read at every upload so that a host can load one SRS with and one without tables (bench.py does).
struct Opts {
    int host_threads_max = 0;        // KZG_HOST_THREADS_MAX: cap of the host pool (transcripts of batch verification); 0 = min(48, the cgroup's CPU quota)
    int host_threads = 0;            // KZG_HOST_THREADS: exactly that many pool threads (measurements)
    int vb_trace = 0;                // KZG_VB_TRACE: 1 = phase times of batch verification on stderr, 2 = per upload chunk
    size_t vb_group_bytes = 0;       // KZG_VB_GROUP_BYTES: packed blob bytes per GPU round of batch verification (default 256 MiB)
    size_t vb_chunk_bytes = 0;       // KZG_VB_CHUNK_BYTES: bytes per upload chunk inside a round (default 16 MiB)
    bool roctx = false;              // KZG_ROCTX: roctx ranges around the host phases (rocprofv3 --marker-trace)
    double exchange_timeout_s = 60;  // KZG_EXCHANGE_TIMEOUT_S: bound of the wait for a collective of the _rccl entries
    const char* rccl_lib = nullptr;  // KZG_RCCL_LIB: the RCCL to dlopen instead of the one already in the process
    int ntt_tile_log = 0;            // KZG_NTT_TILE_LOG: 10 / 11 = one NTT tile size at every transform size (0: by size)
};
const Opts& opts();
bool opt_no_precompute();            // KZG_NO_PRECOMPUTE=1: SRS uploads build no window tables (and no per-bit tables): 64 B per point
bool opt_no_naf();                   // KZG_NO_NAF=1: SRS uploads build no per-bit tables (16 KiB per point saved; fixed-window MSMs)

// process-wide caches keyed by device (NTT twiddles, g1_ifft scalar sets): released when the LAST context of a device is destroyed
void ntt_release_device_caches(int dev);
void g1fft_release_device_caches(int dev);

}  // namespace kzg

// ---- roctx phase ranges (tracing) ------------------------------------------------------------------------
// KZG_ROCTX=1 makes the library mark its host-side phases (enqueue of the sort / accumulate / reduction kernels of an MSM, the host
// epilogue, NTT, proof pipeline, the stages of batch verification) with roctx ranges, visible in `rocprofv3 --marker-trace`.  The
// marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is loaded with dlopen on first use: no link-time dependency,
// no cost when the variable is unset.
namespace kzg {
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
struct RoctxPhases {             // consecutive phases of one function; whatever is open closes on every return path
    bool open = false;
    void begin(const char* name) { end(); roctx_push(name); open = true; }
    void end() { if (open) { roctx_pop(); open = false; } }
    ~RoctxPhases() { end(); }
};
}  // namespace kzg

#define KZG_HIP_TRY(ctx, expr)                                                     \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return kzg::set_error((ctx), _e, #expr);             \
    } while (0)
