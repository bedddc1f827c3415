// capi.hip — the C-ABI of include/kzg_bn254_mi355x.h.  Plain pointers and sizes only.
#include "engine.h"
#include "host_curve.h"
#include "host_pairing.h"
#include "host_sha256.h"
#include "host_transcript.h"
#include "host_lagrange.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <new>
#include <pthread.h>
#include <thread>
#include <vector>

namespace kzg {

// roctx ranges through dlopen (engine.h)
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        if (!opts().roctx) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const RoctxApi& roctx_api() { static const RoctxApi api; return api; }
}  // namespace
void roctx_push(const char* name) { const RoctxApi& a = roctx_api(); if (a.push) (void)a.push(name); }
void roctx_pop() { const RoctxApi& a = roctx_api(); if (a.pop) (void)a.pop(); }

// ---- the environment variables of the library, all of them (engine.h Opts; documented in the header and in INTEGRATION.md) ------------------
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }
const Opts& opts() {
    static const Opts o = []() {
        Opts v;
        v.host_threads_max = env_int("KZG_HOST_THREADS_MAX", 0);
        v.host_threads = env_int("KZG_HOST_THREADS", 0);
        v.vb_trace = env_int("KZG_VB_TRACE", 0);
        { const char* e = getenv("KZG_VB_GROUP_BYTES"); v.vb_group_bytes = e && atol(e) > 0 ? (size_t)atol(e) : 0; }
        { const char* e = getenv("KZG_VB_CHUNK_BYTES"); v.vb_chunk_bytes = e && atol(e) > 0 ? (size_t)atol(e) : 0; }
        v.roctx = env_int("KZG_ROCTX", 0) != 0;
        { const char* e = getenv("KZG_EXCHANGE_TIMEOUT_S"); const double t = e ? atof(e) : 60.0; v.exchange_timeout_s = t > 0 ? t : 60.0; }
        { const char* e = getenv("KZG_RCCL_LIB"); v.rccl_lib = e && *e ? e : nullptr; }
        v.ntt_tile_log = env_int("KZG_NTT_TILE_LOG", 0);
        return v;
    }();
    return o;
}
bool opt_no_precompute() { return env_int("KZG_NO_PRECOMPUTE", 0) != 0; }
bool opt_no_naf() { return env_int("KZG_NO_NAF", 0) != 0; }

int32_t set_error(kzg_ctx* ctx, hipError_t e, const char* where) {
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
        ctx->last_error = buf;
    }
    (void)hipGetLastError();
    return KZG_ERR_DEVICE;
}

// polynomial pipeline (poly.hip)
int32_t proof_run(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4],
                  uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y, bool want_proof, size_t coeff_lo, uint64_t* out_xyzz);
int32_t proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4], int slot, const void* d_resident = nullptr);
int32_t proof_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y);
int32_t roots_run(kzg_ctx* ctx, uint64_t* out, size_t n);
// Lagrange-sharded proofs (lagrange.hip)
int32_t lag_begin(kzg_ctx* ctx, const kzg_srs* shard, size_t base, const void* evals, bool on_device, size_t len, size_t n, const uint64_t z[4], int slot, int commit_slot);
int32_t lag_partial_y(kzg_ctx* ctx, int slot, uint64_t out[8]);
int32_t lag_continue(kzg_ctx* ctx, int slot, const uint64_t y[4]);
int32_t lag_end(kzg_ctx* ctx, int slot, uint64_t out_part[32], uint64_t* out_commit);
void lag_abort(kzg_ctx* ctx, int slot);
int32_t lag_quotient_eval_on_domain(kzg_ctx* ctx, const uint64_t z[4], const uint64_t* evals, size_t n, const uint64_t value[4], uint64_t out[4]);
// (lag_fold_y / lag_fold_proof: host_lagrange.h)
int32_t vb_evaluate_setup(kzg_ctx* ctx, size_t packed_len, size_t nb);
int32_t vb_evaluate_enqueue(kzg_ctx* ctx, const uint8_t* packed, const void* meta_host, size_t nb, size_t b0, size_t b1, const uint64_t* zs,
                            uint8_t* small_pinned);
int32_t vb_evaluate_finish(kzg_ctx* ctx, size_t nb, uint64_t* ys_out, uint8_t* fallback_out);
int32_t blob_to_fr_run(kzg_ctx* ctx, const uint8_t* bytes, size_t len, size_t n_padded, void** d_out,
                       hipStream_t st = nullptr, DeviceBuffer* d_bytes = nullptr, DeviceBuffer* d_elems = nullptr);

}  // namespace kzg

using namespace kzg;

extern "C" {

const char* kzg_status_message(int32_t status) {
    switch (status) {
        case KZG_OK: return "ok";
        case KZG_ERR_INVALID_ARG: return "invalid argument";
        case KZG_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
        case KZG_ERR_DEVICE: return "HIP runtime error";
        case KZG_ERR_MSM_LENGTH_MISMATCH: return "MSM Error: bases and scalars have different lengths";
        case KZG_ERR_SRS_CAPACITY_EXCEEDED: return "SRS capacity exceeded";
        case KZG_ERR_POLY_LENGTH: return "polynomial length is not correct";
        case KZG_ERR_NOT_POWER_OF_TWO: return "length provided is not a power of 2";
        case KZG_ERR_DOMAIN: return "Could not perform IFFT due to domain consturction error";
        case KZG_ERR_ROOTS_LENGTH: return "inconsistent length between blob and root of unities";
        case KZG_ERR_INVALID_INPUT_LENGTH: return "Invalid input length";
        case KZG_ERR_TOO_LARGE: return "Input size exceeds maximum polynomial size";
        case KZG_ERR_ROOT_NOT_FOUND: return "Root of unity not found";
        case KZG_ERR_ZERO_LENGTH: return "Length of data after padding is 0";
        case KZG_ERR_SRS_LENGTH: return "the length of data after padding is not valid with respect to the SRS";
        case KZG_ERR_DESERIALIZE: return "point at infinity not coded properly for g1";
        case KZG_ERR_NOT_ON_CURVE: return "compressed g1 point not on curve";
        case KZG_ERR_G1_NOT_ON_CURVE: return "G1 point not on curve";
        case KZG_ERR_G2_TAU_NOT_ON_CURVE: return "Invalid trusted setup: G2_TAU not on curve";
        case KZG_ERR_TAU_EQUALS_Z: return "Evaluation point equals trusted setup secret";
        case KZG_ERR_PEER: return "another rank of the communicator failed in this collective call";
        case KZG_ERR_EXCHANGE_TIMEOUT: return "the exchange between the ranks timed out";
        case KZG_ERR_IO: return "file could not be read or written";
        default: return "unknown status";
    }
}

int32_t kzg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// live contexts per device: the caches of ntt.hip / g1fft.hip outlive a context (they are shared by the contexts of a device) and are
// freed with the last one (they were never freed)
static std::mutex g_ctx_count_mu;
static std::map<int, int> g_ctx_count;

int32_t kzg_ctx_create(int32_t device_id, kzg_ctx** out) {
    if (!out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    int n = kzg_device_count();
    if (n <= 0 || device_id < 0 || device_id >= n) return KZG_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) { (void)hipGetLastError(); return KZG_ERR_NO_DEVICE; }
    kzg_ctx* ctx = new (std::nothrow) kzg_ctx();
    if (!ctx) return KZG_ERR_INVALID_ARG;
    ctx->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            ctx->acc_wave_slots = (uint32_t)cus * 4u * 3u;                // k_msm_accumulate: 3 waves per SIMD (KZG_ACC_WAVES)
        else (void)hipGetLastError();
    }
    {
        std::lock_guard<std::mutex> lk(g_ctx_count_mu);          // (taken before the handle exists: no call on it can reach a cache that kzg_ctx_destroy is freeing)
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); delete ctx; return KZG_ERR_DEVICE; }
        ++g_ctx_count[device_id];
    }
    *out = ctx;
    return KZG_OK;
}

void kzg_ctx_destroy(kzg_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    blob_stream_release(ctx);                       // first: joins the transcript threads of blob jobs still in flight (they may enqueue work until then)
    (void)hipStreamSynchronize(ctx->stream);
    for (auto st : ctx->stream_x) if (st) (void)hipStreamSynchronize(st);
    if (ctx->lag_stream) (void)hipStreamSynchronize(ctx->lag_stream);
    msm_drop_slots(ctx);
    ctx->last_sorted = nullptr; ctx->last_sorted_stream = nullptr;
    ctx->msm.release();
    for (auto& w : ctx->msm_x) w.release();
    for (auto& st : ctx->stream_x) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
    ctx->ntt.release();
    for (auto& w : ctx->ntt_x) w.release();
    for (auto& ps : ctx->poly) ps.release();
    ctx->rccl_buf.release();
    if (ctx->rccl_pinned) { (void)hipHostFree(ctx->rccl_pinned); ctx->rccl_pinned = nullptr; ctx->rccl_pinned_bytes = 0; }
    if (ctx->vb_pinned) { (void)hipHostFree(ctx->vb_pinned); ctx->vb_pinned = nullptr; ctx->vb_pinned_bytes = 0; }
    for (auto& t : ctx->ondomain_inv) if (t) { (void)hipFree(t); t = nullptr; }
    for (auto& ev : ctx->lag_uploaded) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    for (auto& ev : ctx->lag_phase1) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (ctx->lag_stream) { (void)hipStreamSynchronize(ctx->lag_stream); (void)hipStreamDestroy(ctx->lag_stream); ctx->lag_stream = nullptr; }
    (void)hipStreamDestroy(ctx->stream);
    const int dev = ctx->device;
    delete ctx;
    // the caches are released UNDER the count mutex: a context created on this device meanwhile waits in kzg_ctx_create until they are gone
    // and then builds its own, instead of picking up pointers that are about to be freed
    std::lock_guard<std::mutex> lk(g_ctx_count_mu);
    const bool last = --g_ctx_count[dev] <= 0;
    if (last) { g_ctx_count.erase(dev); kzg::ntt_release_device_caches(dev); kzg::g1fft_release_device_caches(dev); }
}

const char* kzg_ctx_last_error(const kzg_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int32_t kzg_ctx_set_msm_window(kzg_ctx* ctx, int32_t c_bits, int32_t segment_len) {
    if (!ctx || c_bits < 0 || c_bits > 16 || c_bits == 1 || segment_len < 0) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->msm_c_override = c_bits;
    ctx->msm_seg_override = segment_len;
    return KZG_OK;
}

int32_t kzg_ctx_set_reduction_lanes(kzg_ctx* ctx, int32_t lanes) {
    if (!ctx || (lanes != 0 && lanes != 2 && lanes != 4)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->reduction_lanes = lanes;
    return KZG_OK;
}

int32_t kzg_ctx_set_profiling(kzg_ctx* ctx, int32_t enable) {
    if (!ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable != 0;
    for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl) {
        MsmWorkspace& w = ctx->slot_msm(sl);
        for (auto& v : w.phase_ms) v = 0;
        w.profiled_launches = 0;
        w.profiled_pairs = 0;
        w.profiled_entries = 0;
    }
    return KZG_OK;
}

int32_t kzg_ctx_get_msm_profile(kzg_ctx* ctx, double phase_ms_out[8], uint64_t* launches, uint64_t* pairs) {
    if (!ctx || !phase_ms_out) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 0; i < 8; ++i) phase_ms_out[i] = 0;
    uint64_t nl = 0, np = 0;
    for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl) {
        MsmWorkspace& w = ctx->slot_msm(sl);
        for (int i = 0; i < 8; ++i) phase_ms_out[i] += w.phase_ms[i];
        nl += w.profiled_launches;
        np += w.profiled_pairs;
    }
    if (launches) *launches = nl;
    if (pairs) *pairs = np;
    return KZG_OK;
}
int32_t kzg_ctx_get_msm_profile_entries(kzg_ctx* ctx, uint64_t* entries) {
    if (!ctx || !entries) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint64_t ne = 0;
    for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl) ne += ctx->slot_msm(sl).profiled_entries;
    *entries = ne;
    return KZG_OK;
}

// ---- SRS ------------------------------------------------------------------------------------------
// off_curve != nullptr: the points are also validated on the device (y^2 == x^3 + 3 or identity); *off_curve = 1 if one fails
static int32_t upload_points(kzg_ctx* ctx, const uint64_t* xy, size_t n, uint4* d_out, DeviceBuffer& staging, uint32_t* off_curve = nullptr) {
    KZG_HIP_TRY(ctx, staging.reserve(n * 64 + 64));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(staging.p, xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* d_flag = nullptr;
    if (off_curve) {
        d_flag = reinterpret_cast<uint32_t*>(static_cast<char*>(staging.p) + n * 64);
        KZG_HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
    }
    int32_t rc = points_wire_to_device(ctx, staging.as<uint4>(), d_out, n, d_flag);
    if (rc != KZG_OK) return rc;
    if (off_curve) KZG_HIP_TRY(ctx, hipMemcpyAsync(off_curve, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

static int32_t srs_upload_impl(kzg_ctx* ctx, const uint64_t* g1_xy_mont, size_t n_points, kzg_srs** out, bool tables) {
    if (!ctx || !out || (!g1_xy_mont && n_points)) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_points > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    kzg_srs* s = new (std::nothrow) kzg_srs();
    if (!s) return KZG_ERR_INVALID_ARG;
    s->ctx = ctx;
    s->n = n_points;
    if (n_points) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_points), n_points * 64);
        if (e != hipSuccess) { delete s; return set_error(ctx, e, "hipMalloc(srs)"); }
        int32_t rc = upload_points(ctx, g1_xy_mont, n_points, s->d_points, ctx->msm.bases_wire);
        if (rc == KZG_OK && tables) rc = srs_precompute(ctx, s);
        if (rc != KZG_OK) { (void)hipFree(s->d_points); delete s; return rc; }
    }
    *out = s;
    return KZG_OK;
}
int32_t kzg_srs_upload(kzg_ctx* ctx, const uint64_t* g1_xy_mont, size_t n_points, kzg_srs** out) {
    return srs_upload_impl(ctx, g1_xy_mont, n_points, out, true);
}

int32_t kzg_srs_generate(kzg_ctx* ctx, const uint64_t tau_mont[4], uint64_t first_power, size_t n_points, kzg_srs** out) {
    if (!ctx || !out || !tau_mont) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_points > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    kzg_srs* s = new (std::nothrow) kzg_srs();
    if (!s) return KZG_ERR_INVALID_ARG;
    s->ctx = ctx;
    s->n = n_points;
    if (n_points) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_points), n_points * 64);
        if (e != hipSuccess) { delete s; return set_error(ctx, e, "hipMalloc(srs)"); }
        int32_t rc = srs_generate(ctx, tau_mont, first_power, n_points, s->d_points);
        if (rc == KZG_OK) rc = srs_precompute(ctx, s);
        if (rc != KZG_OK) { (void)hipFree(s->d_points); delete s; return rc; }
    }
    *out = s;
    return KZG_OK;
}

static int32_t srs_load_compressed(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index, bool ark_le);
int32_t kzg_srs_load_compressed_be(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index) {
    return srs_load_compressed(ctx, bytes, n_points, out, bad_index, false);
}
int32_t kzg_srs_load_compressed_ark_le(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index) {
    return srs_load_compressed(ctx, bytes, n_points, out, bad_index, true);
}
static int32_t srs_load_compressed(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index, bool ark_le) {
    if (!ctx || !out || (n_points && !bytes)) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_points > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    kzg_srs* s = new (std::nothrow) kzg_srs();
    if (!s) return KZG_ERR_INVALID_ARG;
    s->ctx = ctx;
    s->n = n_points;
    if (n_points) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_points), n_points * 64);
        if (e != hipSuccess) { delete s; return set_error(ctx, e, "hipMalloc(srs)"); }
        uint32_t kind = 0, idx = 0;
        int32_t rc = srs_decompress(ctx, bytes, n_points, s->d_points, &kind, &idx, ark_le);
        if (rc == KZG_OK && kind != 0) {
            if (bad_index) *bad_index = idx;
            rc = kind == 1 ? KZG_ERR_DESERIALIZE : KZG_ERR_NOT_ON_CURVE;
        }
        if (rc == KZG_OK) rc = srs_precompute(ctx, s);
        if (rc != KZG_OK) { (void)hipFree(s->d_points); delete s; return rc; }
    }
    *out = s;
    return KZG_OK;
}

// ---- packed SRS file: the decoded points as they cross the C-ABI (64 B each), so that a restart skips the decoding of the ceremony file ----
// layout: "KZGSRS1\0" | u64 n | u64 0 | SHA-256 of the payload (32 B) | payload = n x 64 B (x || y Montgomery words, identity = zeros), little-endian
namespace {
const char PACKED_MAGIC[8] = {'K', 'Z', 'G', 'S', 'R', 'S', '1', 0};
constexpr size_t PACKED_HEADER = 8 + 8 + 8 + 32;
void sha256_of(const uint8_t* data, size_t len, uint8_t out[32]) {
    kzg_host::Sha256 sh;
    kzg_host::sha256_init(sh);
    kzg_host::sha256_update(sh, data, len);
    kzg_host::sha256_final(sh, out);
}
}  // namespace

int32_t kzg_srs_save_packed(kzg_ctx* ctx, const kzg_srs* srs, const char* path) {
    if (!ctx || !srs || srs->ctx != ctx || !path) return KZG_ERR_INVALID_ARG;
    std::vector<uint64_t> pts(srs->n * 8 + 1);
    if (srs->n) { const int32_t rc = kzg_srs_download(ctx, srs, 0, srs->n, pts.data()); if (rc != KZG_OK) return rc; }
    uint8_t head[PACKED_HEADER] = {0};
    memcpy(head, PACKED_MAGIC, 8);
    const uint64_t n64 = (uint64_t)srs->n;
    memcpy(head + 8, &n64, 8);
    sha256_of(reinterpret_cast<const uint8_t*>(pts.data()), srs->n * 64, head + 24);
    FILE* f = fopen(path, "wb");
    if (!f) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = std::string("cannot create ") + path; return KZG_ERR_IO; }
    const bool ok = fwrite(head, 1, sizeof head, f) == sizeof head && (srs->n == 0 || fwrite(pts.data(), 64, srs->n, f) == srs->n);
    const bool closed = fclose(f) == 0;
    if (!ok || !closed) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = std::string("short write to ") + path; return KZG_ERR_IO; }
    return KZG_OK;
}

int32_t kzg_srs_load_packed(kzg_ctx* ctx, const char* path, size_t points_to_load, kzg_srs** out) {
    if (!ctx || !path || !out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = std::string("cannot open ") + path; return KZG_ERR_IO; }
    uint8_t head[PACKED_HEADER];
    uint64_t n64 = 0;
    std::vector<uint64_t> pts;
    int32_t rc = KZG_OK;
    if (fread(head, 1, sizeof head, f) != sizeof head || memcmp(head, PACKED_MAGIC, 8) != 0) rc = KZG_ERR_DESERIALIZE;
    if (rc == KZG_OK) {
        memcpy(&n64, head + 8, 8);
        if (n64 > ((uint64_t)1 << 28)) rc = KZG_ERR_TOO_LARGE;
    }
    if (rc == KZG_OK) {                                            // the header's count is believed only if the file has exactly that many bytes
        const long here = ftell(f);
        long size = -1;
        if (here >= 0 && fseek(f, 0, SEEK_END) == 0) { size = ftell(f); if (fseek(f, here, SEEK_SET) != 0) size = -1; }
        if (size < 0 || (uint64_t)size != (uint64_t)PACKED_HEADER + n64 * 64) rc = KZG_ERR_DESERIALIZE;   // truncated, or bytes behind the payload
    }
    if (rc == KZG_OK) {
        try { pts.resize((size_t)n64 * 8 + 1); } catch (const std::bad_alloc&) { rc = KZG_ERR_TOO_LARGE; }
        if (rc == KZG_OK && n64 && fread(pts.data(), 64, (size_t)n64, f) != (size_t)n64) rc = KZG_ERR_DESERIALIZE;
    }
    fclose(f);
    if (rc == KZG_OK) {
        uint8_t dig[32];
        sha256_of(reinterpret_cast<const uint8_t*>(pts.data()), (size_t)n64 * 64, dig);
        if (memcmp(dig, head + 24, 32) != 0) rc = KZG_ERR_DESERIALIZE;
    }
    if (rc != KZG_OK) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = std::string(path) + ": not a packed SRS file of this library, or damaged"; return rc; }
    const size_t n = points_to_load ? points_to_load : (size_t)n64;
    if (n > (size_t)n64) return KZG_ERR_SRS_LENGTH;               // more points asked for than the file holds
    // the points are validated on the device while they are converted (y^2 = x^3 + 3 or the identity): a file from elsewhere cannot smuggle in garbage
    kzg_srs* s = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
        s = new (std::nothrow) kzg_srs();
        if (!s) return KZG_ERR_INVALID_ARG;
        s->ctx = ctx;
        s->n = n;
        if (n) {
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_points), n * 64);
            if (e != hipSuccess) { delete s; return set_error(ctx, e, "hipMalloc(srs)"); }
            uint32_t off_curve = 0;
            rc = upload_points(ctx, pts.data(), n, s->d_points, ctx->msm.bases_wire, &off_curve);
            if (rc == KZG_OK && off_curve) rc = KZG_ERR_NOT_ON_CURVE;
            if (rc == KZG_OK) rc = srs_precompute(ctx, s);
            if (rc != KZG_OK) { (void)hipFree(s->d_points); delete s; return rc; }
        }
    }
    *out = s;
    return KZG_OK;
}

int32_t kzg_srs_download(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, size_t n, uint64_t* out_xy_mont) {
    if (!ctx || !srs || srs->ctx != ctx || (n && !out_xy_mont)) return KZG_ERR_INVALID_ARG;
    if (offset > srs->n || n > srs->n - offset) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return srs_download(ctx, srs->d_points + 4 * offset, n, out_xy_mont);
}

void kzg_srs_free(kzg_srs* srs) {
    if (!srs) return;
    for (auto& kv : srs->lagrange) kzg_srs_free(kv.second);
    srs->lagrange.clear();
    if (srs->d_points) { (void)hipSetDevice(srs->ctx->device); (void)hipFree(srs->d_points); }
    if (srs->d_small) { (void)hipSetDevice(srs->ctx->device); (void)hipFree(srs->d_small); }
    if (srs->d_bits) { (void)hipSetDevice(srs->ctx->device); (void)hipFree(srs->d_bits); }
    if (srs->d_t3) { (void)hipSetDevice(srs->ctx->device); (void)hipFree(srs->d_t3); }
    delete srs;
}

size_t kzg_srs_len(const kzg_srs* srs) { return srs ? srs->n : 0; }
int32_t kzg_srs_has_bit_tables(kzg_srs* srs, int32_t build) {
    if (!srs) return 0;
    if (srs_bits(srs)) return 1;
    if (!build || !srs->ctx) return 0;
    std::lock_guard<std::mutex> lk(srs->ctx->mu);
    if (hipSetDevice(srs->ctx->device) != hipSuccess) return 0;
    (void)srs_build_bit_tables(srs->ctx, srs, true);
    return srs_bits(srs) ? 1 : 0;
}

// Lagrange basis of the first n points as an SRS of its own (device resident, with its window tables); len < n: only the points
// [lo, lo + len) of it are kept (a rank's shard of the basis, kzg_srs_lagrange_shard)
static int32_t build_lagrange(kzg_ctx* ctx, const kzg_srs* srs, size_t n, kzg_srs** out, size_t lo = 0, size_t len = (size_t)-1) {
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;             // kzg.rs:265-269
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;                               // kzg.rs:275-278
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    if (len == (size_t)-1) len = n;
    kzg_srs* s = new (std::nothrow) kzg_srs();
    if (!s) return KZG_ERR_INVALID_ARG;
    s->ctx = ctx;
    s->n = len;
    s->lagrange_of = n;
    uint4* full = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&full), n * 64);
    if (e != hipSuccess) { delete s; return set_error(ctx, e, "hipMalloc(lagrange srs)"); }
    int32_t rc = g1_ifft_device(ctx, srs, n, full, false);
    if (rc == KZG_OK) { hipError_t e2 = hipStreamSynchronize(ctx->stream); if (e2 != hipSuccess) rc = set_error(ctx, e2, "g1_ifft"); }
    if (rc == KZG_OK && len != n) {                                                 // keep the slice only
        uint4* part = nullptr;
        if (len) {
            e = hipMalloc(reinterpret_cast<void**>(&part), len * 64);
            if (e == hipSuccess) e = hipMemcpy(part, full + 4 * lo, len * 64, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { if (part) (void)hipFree(part); part = nullptr; rc = set_error(ctx, e, "lagrange shard"); }
        }
        (void)hipFree(full);
        full = part;
    }
    s->d_points = full;
    if (rc == KZG_OK && len) rc = srs_precompute(ctx, s);
    if (rc != KZG_OK) { if (s->d_points) (void)hipFree(s->d_points); delete s; return rc; }
    *out = s;
    return KZG_OK;
}

int32_t kzg_srs_lagrange(kzg_ctx* ctx, const kzg_srs* srs, size_t n, kzg_srs** out) {
    if (!ctx || !srs || srs->ctx != ctx || !out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return build_lagrange(ctx, srs, n, out);
}

int32_t kzg_srs_lagrange_shard(kzg_ctx* ctx, const kzg_srs* srs, size_t n, size_t lo, size_t len, kzg_srs** out) {
    if (!ctx || !srs || srs->ctx != ctx || !out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (lo > n || len > n - lo) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return build_lagrange(ctx, srs, n, out, lo, len);
}

int32_t kzg_srs_slice(kzg_ctx* ctx, const kzg_srs* srs, size_t lo, size_t len, kzg_srs** out) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (lo > srs->n || len > srs->n - lo) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    kzg_srs* s = new (std::nothrow) kzg_srs();
    if (!s) return KZG_ERR_INVALID_ARG;
    s->ctx = ctx;
    s->n = len;
    s->lagrange_of = srs->lagrange_of;
    if (len) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_points), len * 64);
        if (e == hipSuccess) e = hipMemcpy(s->d_points, srs->d_points + 4 * lo, len * 64, hipMemcpyDeviceToDevice);   // table 0 of `srs` = its points
        if (e != hipSuccess) { if (s->d_points) (void)hipFree(s->d_points); delete s; return set_error(ctx, e, "kzg_srs_slice"); }
        int32_t rc = srs_precompute(ctx, s);
        if (rc != KZG_OK) { (void)hipFree(s->d_points); delete s; return rc; }
    }
    *out = s;
    return KZG_OK;
}

int32_t kzg_srs_cache_lagrange(kzg_ctx* ctx, kzg_srs* srs, size_t n) {
    if (!ctx || !srs || srs->ctx != ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (srs_cached_lagrange(srs, n)) return KZG_OK;
    kzg_srs* l = nullptr;
    int32_t rc = build_lagrange(ctx, srs, n, &l);           // (under ctx->mu: one builder at a time; complete and synchronised on return)
    if (rc != KZG_OK) return rc;
    std::lock_guard<std::mutex> lazy(srs->lazy_mu);
    srs->lagrange[n] = l;
    return KZG_OK;
}

int32_t kzg_srs_drop_lagrange(kzg_ctx* ctx, kzg_srs* srs) {
    if (!ctx || !srs || srs->ctx != ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::lock_guard<std::mutex> lazy(srs->lazy_mu);
    for (auto& kv : srs->lagrange) kzg_srs_free(kv.second);
    srs->lagrange.clear();
    return KZG_OK;
}


// ---- MSM ------------------------------------------------------------------------------------------
static int32_t stage_scalars(kzg_ctx* ctx, const uint64_t* scalars, size_t n, const void** d_out) {
    KZG_HIP_TRY(ctx, ctx->msm.scalars.reserve(n * 32 + 32));
    if (n) KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->msm.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    *d_out = ctx->msm.scalars.p;
    return KZG_OK;
}
static void write_identity(uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    if (out_xy) memset(out_xy, 0, 64);
    if (out_inf) *out_inf = 1;
    if (out_xyzz) memset(out_xyzz, 0, 128);
}

int32_t kzg_msm_g1(kzg_ctx* ctx, const uint64_t* bases_xy_mont, size_t n_bases, const uint64_t* scalars_mont, size_t n_scalars,
                   uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !out_xy_mont) return KZG_ERR_INVALID_ARG;
    if (n_bases != n_scalars) return KZG_ERR_MSM_LENGTH_MISMATCH;
    if (n_bases && (!bases_xy_mont || !scalars_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_bases == 0) { write_identity(out_xy_mont, out_is_infinity, nullptr); return KZG_OK; }
    KZG_HIP_TRY(ctx, ctx->msm.bases.reserve(n_bases * 64));
    int32_t rc = upload_points(ctx, bases_xy_mont, n_bases, ctx->msm.bases.as<uint4>(), ctx->msm.bases_wire);
    if (rc != KZG_OK) return rc;
    const void* d_scalars;
    rc = stage_scalars(ctx, scalars_mont, n_scalars, &d_scalars);
    if (rc != KZG_OK) return rc;
    MsmBases b;
    b.points = ctx->msm.bases.as<uint4>();
    return msm_run(ctx, b, d_scalars, n_bases, out_xy_mont, out_is_infinity, nullptr);
}

static int32_t msm_g1_batch_impl(kzg_ctx* ctx, const uint64_t* bases_xy_mont, const uint64_t* scalars_mont, size_t n, size_t batch,
                                 uint64_t* out_xy_mont, uint8_t* out_is_infinity, uint32_t* off_curve) {
    if (!ctx || !out_xy_mont || batch == 0 || batch > 64) return KZG_ERR_INVALID_ARG;
    if (n && (!bases_xy_mont || !scalars_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (off_curve) *off_curve = 0;
    if (n == 0) {
        for (size_t i = 0; i < batch; ++i) write_identity(out_xy_mont + 8 * i, out_is_infinity ? out_is_infinity + i : nullptr, nullptr);
        return KZG_OK;
    }
    const size_t total = n * batch;
    KZG_HIP_TRY(ctx, ctx->msm.bases.reserve(total * 64));
    int32_t rc = upload_points(ctx, bases_xy_mont, total, ctx->msm.bases.as<uint4>(), ctx->msm.bases_wire, off_curve);
    if (rc != KZG_OK) return rc;
    if (off_curve && *off_curve) return KZG_OK;              // the caller reports the invalid point; no MSM over it
    const void* d_scalars;
    rc = stage_scalars(ctx, scalars_mont, total, &d_scalars);
    if (rc != KZG_OK) return rc;
    return msm_run_batch(ctx, ctx->msm.bases.as<uint4>(), d_scalars, n, (uint32_t)batch, out_xy_mont, out_is_infinity);
}
int32_t kzg_msm_g1_batch(kzg_ctx* ctx, const uint64_t* bases_xy_mont, const uint64_t* scalars_mont, size_t n, size_t batch,
                         uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    return msm_g1_batch_impl(ctx, bases_xy_mont, scalars_mont, n, batch, out_xy_mont, out_is_infinity, nullptr);
}

constexpr size_t MSM_SPLIT_MAX = (size_t)1 << 21;     // halves of up to 2^20 pairs: one launch each

// (the caller holds ctx->mu)
static int32_t msm_srs_locked(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const void* scalars, bool on_device, size_t n,
                              uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || (!out_xy && !out_xyzz)) return KZG_ERR_INVALID_ARG;   // an SRS may be shared by the contexts of its GPU
    if (n && !scalars) return KZG_ERR_INVALID_ARG;
    if (offset > srs->n || n > srs->n - offset) return KZG_ERR_MSM_LENGTH_MISMATCH;
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) { write_identity(out_xy, out_inf, out_xyzz); return KZG_OK; }
    const void* d_scalars = scalars;
    if (!on_device) {
        // A large MSM from a HOST buffer, one call at a time: as TWO halves on two slots.  The upload of the second half (a pageable copy: this
        // thread sits in it) runs beside the kernels of the first half, and the first half's accumulate beside the second half's sort; the two
        // partial sums are added on the host.  Resident, two halves cost what the whole costs (tools/probe_split_lone.py: 2^19 0.872 against
        // 0.887 ms, 2^20 1.444 against 1.445) -- so the hidden part of the upload is the gain: 2^19 1.15 -> 1.07 ms, 2^20 2.00 -> 1.76.
        // Below 2^19 pairs the second bucket set costs more than the hidden copy (2^18: 0.601 against 0.553 resident).
        // share of the FIRST half (the smaller it is, the sooner the GPU starts and the more of the upload is hidden; too small and the second MSM runs alone):
        // measured from host buffers, 0.375 / 0.44 / 0.5 / 0.56: 2^19 1.116 / 1.070 / 1.084 / 1.111 ms (unsplit 1.148), 2^20 1.756 / 1.786 / 1.901 / 1.863 (unsplit 2.000)
        const double split_frac = n >= ((size_t)1 << 20) ? 0.375 : 0.44;
        const bool idle = !ctx->slot_pending[0] && !ctx->slot_pending[1] && ctx->lag[0].phase == 0 && ctx->lag[1].phase == 0;
        if (idle && n >= ((size_t)1 << 19) && n <= MSM_SPLIT_MAX && ctx->msm_c_override == 0 && srs_bits(srs)) {
            const size_t half = ((size_t)((double)n * split_frac) + 255) / 256 * 256;
            const uint64_t* sc = static_cast<const uint64_t*>(scalars);
            const size_t lo[2] = {0, half}, len[2] = {half, n - half};
            int32_t rc = KZG_OK;
            int begun = 0;
            for (int h = 0; h < 2 && rc == KZG_OK; ++h) {
                hipStream_t st = nullptr;
                rc = msm_slot_stream(ctx, h, &st);
                if (rc != KZG_OK) break;
                MsmWorkspace& ws = ctx->slot_msm(h);
                hipError_t e = ws.scalars.reserve(len[h] * 32 + 32);
                if (e == hipSuccess) e = hipMemcpyAsync(ws.scalars.p, sc + 4 * lo[h], len[h] * 32, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) { rc = set_error(ctx, e, "split upload"); break; }
                rc = msm_begin(ctx, h, srs_bases(srs, offset + lo[h], len[h], true), ws.scalars.p, len[h]);
                if (rc == KZG_OK) ++begun;
            }
            kzg_host::Xyzz parts[2] = {kzg_host::xyzz_inf(), kzg_host::xyzz_inf()};
            for (int h = 0; h < begun; ++h) {                          // (also after a failure of the second half: nothing may stay in flight)
                uint64_t w[16];
                const int32_t r2 = msm_end(ctx, h, nullptr, nullptr, w);
                if (r2 == KZG_OK) memcpy(&parts[h], w, 128); else if (rc == KZG_OK) rc = r2;
            }
            if (rc != KZG_OK) return rc;
            const kzg_host::Xyzz total = kzg_host::xyzz_add(parts[0], parts[1]);
            if (out_xyzz) memcpy(out_xyzz, &total, 128);
            if (out_xy) kzg_host::xyzz_to_affine(total, out_xy, out_inf);
            return KZG_OK;
        }
        int32_t rc = stage_scalars(ctx, static_cast<const uint64_t*>(scalars), n, &d_scalars);
        if (rc != KZG_OK) return rc;
    }
    return msm_run(ctx, srs_bases(srs, offset, n, ctx->msm_c_override == 0), d_scalars, n, out_xy, out_inf, out_xyzz);
}
static int32_t msm_srs_common(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const void* scalars, bool on_device, size_t n,
                              uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    if (!ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return msm_srs_locked(ctx, srs, offset, scalars, on_device, n, out_xy, out_inf, out_xyzz);
}

int32_t kzg_msm_g1_srs(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const uint64_t* scalars_mont, size_t n,
                       uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!out_xy_mont) return KZG_ERR_INVALID_ARG;
    return msm_srs_common(ctx, srs, offset, scalars_mont, false, n, out_xy_mont, out_is_infinity, nullptr);
}
int32_t kzg_msm_g1_srs_device(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const void* d_scalars_mont, size_t n,
                              uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!out_xy_mont) return KZG_ERR_INVALID_ARG;
    return msm_srs_common(ctx, srs, offset, d_scalars_mont, true, n, out_xy_mont, out_is_infinity, nullptr);
}
int32_t kzg_msm_g1_srs_partial_device(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const void* d_scalars_mont, size_t n,
                                      uint64_t out_xyzz_mont[16]) {
    if (!out_xyzz_mont) return KZG_ERR_INVALID_ARG;
    return msm_srs_common(ctx, srs, offset, d_scalars_mont, true, n, nullptr, nullptr, out_xyzz_mont);
}
int32_t kzg_msm_g1_srs_partial(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const uint64_t* scalars_mont, size_t n,
                               uint64_t out_xyzz_mont[16]) {
    if (!out_xyzz_mont) return KZG_ERR_INVALID_ARG;
    return msm_srs_common(ctx, srs, offset, scalars_mont, false, n, nullptr, nullptr, out_xyzz_mont);
}

int32_t kzg_msm_g1_srs_device_begin(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const void* d_scalars_mont, size_t n, int32_t slot) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !d_scalars_mont) return KZG_ERR_INVALID_ARG;
    if (offset > srs->n || n > srs->n - offset) return KZG_ERR_MSM_LENGTH_MISMATCH;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return msm_begin(ctx, slot, srs_bases(srs, offset, n, ctx->msm_c_override == 0), d_scalars_mont, n);
}
int32_t kzg_msm_g1_srs_begin(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, const uint64_t* scalars_mont, size_t n, int32_t slot) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !scalars_mont || slot < 0 || slot >= KZG_NUM_SLOTS) return KZG_ERR_INVALID_ARG;
    if (offset > srs->n || n > srs->n - offset) return KZG_ERR_MSM_LENGTH_MISMATCH;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    // H2D copy on the slot's own stream into the slot's own staging buffer: it overlaps the other slot's kernels
    hipStream_t st = nullptr;
    { int32_t rc = msm_slot_stream(ctx, slot, &st); if (rc != KZG_OK) return rc; }
    MsmWorkspace& ws = ctx->slot_msm(slot);
    KZG_HIP_TRY(ctx, ws.scalars.reserve(n * 32 + 32));
    if (n) KZG_HIP_TRY(ctx, hipMemcpyAsync(ws.scalars.p, scalars_mont, n * 32, hipMemcpyHostToDevice, st));
    return msm_begin(ctx, slot, srs_bases(srs, offset, n, ctx->msm_c_override == 0), ws.scalars.p, n);
}
int32_t kzg_msm_g1_srs_end(kzg_ctx* ctx, int32_t slot, uint64_t* out_xy_mont, uint8_t* out_is_infinity, uint64_t* out_xyzz_mont) {
    if (!ctx || (!out_xy_mont && !out_xyzz_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return msm_end(ctx, slot, out_xy_mont, out_is_infinity, out_xyzz_mont);
}

// ---- one batched launch, asynchronous: `count` scalar sets (separate device buffers, n scalars each) against srs[offset .. offset + n)
size_t kzg_msm_batch_capacity(size_t n) { return kzg::msm_batch_capacity(n); }
int32_t kzg_msm_g1_srs_device_begin_batch(kzg_ctx* ctx, kzg_srs* srs, size_t offset, const void* const* d_scalars_mont, size_t n, size_t count, int32_t slot) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !d_scalars_mont) return KZG_ERR_INVALID_ARG;
    if (offset > srs->n || n > srs->n - offset) return KZG_ERR_MSM_LENGTH_MISMATCH;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint4* bits = srs_bits(srs);
    if (!bits) {
        int32_t rc = srs_build_bit_tables(ctx, srs, true);
        if (rc != KZG_OK) return rc;
        bits = srs_bits(srs);
        if (!bits) { ctx->last_error = "no per-bit tables for this SRS (memory, or KZG_NO_NAF)"; return KZG_ERR_INVALID_ARG; }
    }
    MsmBases b;
    b.points = bits + 4 * offset; b.table_stride = (uint32_t)srs->n; b.c = 7; b.W = 255; b.naf = true;
    return msm_begin_batch(ctx, slot, b, d_scalars_mont, n, count);
}
int32_t kzg_msm_g1_srs_end_batch(kzg_ctx* ctx, int32_t slot, size_t count, uint64_t* out_xy_mont, uint8_t* out_is_infinity, uint64_t* out_xyzz_mont) {
    if (!ctx || (!out_xy_mont && !out_xyzz_mont)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return msm_end_batch(ctx, slot, count, out_xy_mont, out_is_infinity, out_xyzz_mont);
}

int32_t kzg_g1_fold_partials(const uint64_t* partials_xyzz_mont, size_t count, uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!out_xy_mont || (count && !partials_xyzz_mont)) return KZG_ERR_INVALID_ARG;
    kzg_host::Xyzz acc = kzg_host::xyzz_inf();
    for (size_t i = 0; i < count; ++i) {
        kzg_host::Xyzz p;
        memcpy(&p, partials_xyzz_mont + 16 * i, 128);
        acc = kzg_host::xyzz_add(acc, p);
    }
    kzg_host::xyzz_to_affine(acc, out_xy_mont, out_is_infinity);
    return KZG_OK;
}

// ---- verifier surface: O(1) host pairing, data-parallel part on the GPU ---------------------------------------------
static bool load_g2_tau(const uint64_t* g2_tau_mont, kzg_host::G2* out) {
    *out = g2_tau_mont ? kzg_host::g2_from_wire(g2_tau_mont) : kzg_host::g2_tau_mainnet();
    return kzg_host::g2_on_curve(*out);
}

int32_t kzg_g2_generator(uint64_t out_g2_mont[16]) {
    if (!out_g2_mont) return KZG_ERR_INVALID_ARG;
    kzg_host::g2_to_wire(kzg_host::g2_generator(), out_g2_mont);
    return KZG_OK;
}
int32_t kzg_g2_tau_mainnet(uint64_t out_g2_mont[16]) {
    if (!out_g2_mont) return KZG_ERR_INVALID_ARG;
    kzg_host::g2_to_wire(kzg_host::g2_tau_mainnet(), out_g2_mont);
    return KZG_OK;
}
int32_t kzg_g2_mul_generator(const uint64_t scalar_mont[4], uint64_t out_g2_mont[16]) {
    if (!scalar_mont || !out_g2_mont) return KZG_ERR_INVALID_ARG;
    uint64_t k[4];
    kzg_host::fr_wire_to_canonical(scalar_mont, k);
    kzg_host::g2_to_wire(kzg_host::g2_mul_generator(k), out_g2_mont);
    return KZG_OK;
}

int32_t kzg_g2_is_on_curve(const uint64_t g2_mont[16], int32_t* out_on_curve) {
    if (!g2_mont || !out_on_curve) return KZG_ERR_INVALID_ARG;
    *out_on_curve = kzg_host::g2_on_curve(kzg_host::g2_from_wire(g2_mont)) ? 1 : 0;
    return KZG_OK;
}

int32_t kzg_validate_g2_point(const uint64_t g2_mont[16], int32_t* out_reason) {
    if (!g2_mont || !out_reason) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    const G2 p = g2_from_wire(g2_mont);
    *out_reason = 0;
    if (!g2_on_curve(p)) { *out_reason = 1; return KZG_OK; }                               // helpers.rs:741-745
    if (p.inf) { *out_reason = 2; return KZG_OK; }                                          // :747-751
    if (!g2_mul(p, FR_MODULUS_WORDS).inf) { *out_reason = 3; return KZG_OK; }                          // :753-757: [r] P = O <=> P in the order-r subgroup of the twist
    const G2 g = g2_generator();
    if (eq(p.x, g.x) && eq(p.y, g.y)) { *out_reason = 4; return KZG_OK; }                   // :759-763
    return KZG_OK;
}

int32_t kzg_pairings_verify(const uint64_t a1_xy_mont[8], const uint64_t a2_g2_mont[16], const uint64_t b1_xy_mont[8],
                            const uint64_t b2_g2_mont[16], int32_t* out_ok) {
    if (!a1_xy_mont || !a2_g2_mont || !b1_xy_mont || !b2_g2_mont || !out_ok) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    G1 a1 = g1_from_wire(a1_xy_mont), b1 = g1_from_wire(b1_xy_mont);
    G2 a2 = g2_from_wire(a2_g2_mont), b2 = g2_from_wire(b2_g2_mont);
    if (!g1_on_curve(a1) || !g1_on_curve(b1)) return KZG_ERR_G1_NOT_ON_CURVE;
    if (!g2_on_curve(a2) || !g2_on_curve(b2)) return KZG_ERR_INVALID_ARG;
    *out_ok = pairings_verify(a1, a2, b1, b2) ? 1 : 0;
    return KZG_OK;
}

int32_t kzg_verify_proof(const uint64_t commitment_xy_mont[8], const uint64_t proof_xy_mont[8], const uint64_t value_mont[4],
                         const uint64_t z_mont[4], const uint64_t* g2_tau_mont, int32_t* out_ok) {
    if (!commitment_xy_mont || !proof_xy_mont || !value_mont || !z_mont || !out_ok) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    G1 commitment = g1_from_wire(commitment_xy_mont), proof = g1_from_wire(proof_xy_mont);
    if (!g1_on_curve(commitment) || !g1_on_curve(proof)) return KZG_ERR_G1_NOT_ON_CURVE;   // verify.rs:18,22 (G1 has cofactor 1)
    G2 g2_tau;
    if (!load_g2_tau(g2_tau_mont, &g2_tau)) return KZG_ERR_G2_TAU_NOT_ON_CURVE;            // verify.rs:29-33
    uint64_t y[4], z[4];
    fr_wire_to_canonical(value_mont, y);
    fr_wire_to_canonical(z_mont, z);
    G1 commit_minus_value = g1_add(commitment, g1_neg(g1_mul_generator(y)));               // verify.rs:37-42 (fixed-base tables)
    G2 x_minus_z = g2_add(g2_tau, g2_neg(g2_mul_generator(z)));                            // verify.rs:46-51
    if (x_minus_z.inf) return KZG_ERR_TAU_EQUALS_Z;                                        // verify.rs:56-60
    *out_ok = pairings_verify(commit_minus_value, g2_generator(), proof, x_minus_z) ? 1 : 0;   // verify.rs:66-71
    return KZG_OK;
}

}  // extern "C"
namespace { void parallel_for(size_t n, const std::function<void(size_t)>& job); }   // the host pool (defined with the batch verifier's front end below)
extern "C" {
// helpers::pairings_verify for the batch verifier: the two Miller loops on the persistent host pool (a parked worker wakes in ~20 us; host_pairing.h's own
// form starts a std::thread per call), then the one final exponentiation.  Same Fq12 values, bit for bit.
static bool pairings_verify_pooled(const kzg_host::G1& a1, const kzg_host::G2& a2, const kzg_host::G1& b1, const kzg_host::G2& b2) {
    using namespace kzg_host;
    G1 ps[2] = {a1, g1_neg(b1)};
    G2 qs[2] = {a2, b2};
    Fq12 f[2];
    bool bad[2] = {false, false};
    parallel_for(2, [&](size_t k) { f[k] = miller_ate_product(ps + k, qs + k, 1, &bad[k]); });
    if (bad[0] || bad[1]) return false;
    return fq12_is_one(final_exponentiation_x(mul(f[0], f[1])));
}
static int32_t verify_batch_core(kzg_ctx* ctx, const uint64_t* commitments_xy_mont, const uint64_t* zs_mont, const uint64_t* ys_mont,
                                 const uint64_t* proofs_xy_mont, const uint64_t* r_powers_mont, size_t n,
                                 const uint64_t* g2_tau_mont, int32_t* out_ok) {
    using namespace kzg_host;
    // batch.rs:203-210 (every commitment and proof on the curve) is checked on the GPU, on the points the MSM uploads anyway
    // (k_points_wire_to_device_checked: 2n curve equations cost the host 0.5 ms at n = 4096); the error order of the reference is
    // kept: a point off the curve is reported before a bad g2_tau (batch.rs:214-216).
    G2 g2_tau;
    const bool g2_ok = load_g2_tau(g2_tau_mont, &g2_tau);
    // scalars of the three linear combinations (batch.rs:228, :245, :246).  sum_i r^i (C_i - [y_i]G) is evaluated as
    // sum_i r^i C_i - [sum_i r^i y_i] G: the same group element with one fixed-base product instead of n.
    std::vector<uint64_t> bases(3 * n * 8), scalars(3 * n * 4);
    uint64_t s[4] = {0, 0, 0, 0};
    {
        // 2 n field products (0.15 ms at n = 4096 on one core): chunks on the host pool, one partial sum per chunk
        const size_t chunks = n >= 1024 ? 32 : 1, per = (n + chunks - 1) / chunks;
        std::vector<uint64_t> partial(4 * chunks, 0);
        auto body = [&](size_t c) {
            uint64_t acc[4] = {0, 0, 0, 0};
            for (size_t i = c * per; i < std::min(n, (c + 1) * per); ++i) {
                uint64_t t[4];
                fr_mul(r_powers_mont + 4 * i, zs_mont + 4 * i, scalars.data() + (n + i) * 4);      // r^i z_i
                fr_mul(r_powers_mont + 4 * i, ys_mont + 4 * i, t);
                fr_add(acc, t, acc);
            }
            memcpy(partial.data() + 4 * c, acc, 32);
        };
        if (chunks > 1) parallel_for(chunks, body); else body(0);
        for (size_t c = 0; c < chunks; ++c) fr_add(s, partial.data() + 4 * c, s);
    }
    if (n) {
        memcpy(bases.data(), proofs_xy_mont, n * 64);
        memcpy(bases.data() + n * 8, proofs_xy_mont, n * 64);
        memcpy(bases.data() + 2 * n * 8, commitments_xy_mont, n * 64);
        memcpy(scalars.data(), r_powers_mont, n * 32);
        memcpy(scalars.data() + 2 * n * 4, r_powers_mont, n * 32);
    }
    uint64_t sums[3 * 8];
    uint8_t infs[3];
    uint32_t off_curve = 0;
    const auto t_c0 = std::chrono::steady_clock::now();
    int32_t rc = msm_g1_batch_impl(ctx, bases.data(), scalars.data(), n, 3, sums, infs, &off_curve);
    const auto t_c1 = std::chrono::steady_clock::now();
    if (rc != KZG_OK) return rc;
    if (off_curve) return KZG_ERR_G1_NOT_ON_CURVE;
    if (!g2_ok) return KZG_ERR_G2_TAU_NOT_ON_CURVE;
    G1 proof_lincomb = g1_from_wire(sums), proof_z_lincomb = g1_from_wire(sums + 8), c_lincomb = g1_from_wire(sums + 16);
    uint64_t s_int[4];
    fr_wire_to_canonical(s, s_int);
    G1 rhs = g1_add(g1_add(c_lincomb, g1_neg(g1_mul_generator(s_int))), proof_z_lincomb);   // batch.rs:249
    const auto t_c2 = std::chrono::steady_clock::now();
    *out_ok = pairings_verify_pooled(proof_lincomb, g2_tau, rhs, g2_generator()) ? 1 : 0;  // batch.rs:253-254
    if (opts().vb_trace) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "  verify_batch_core n=%zu: three MSMs (upload, kernels, host Horner) %.3f ms, [s]G + point sums %.3f ms, pairing check %.3f ms\n", n, ms(t_c0, t_c1), ms(t_c1, t_c2),
                ms(t_c2, std::chrono::steady_clock::now()));
    }
    return KZG_OK;
}

int32_t kzg_verify_kzg_proof_batch(kzg_ctx* ctx, const uint64_t* commitments_xy_mont, const uint64_t* zs_mont, const uint64_t* ys_mont,
                                   const uint64_t* proofs_xy_mont, const uint64_t* r_powers_mont, size_t n,
                                   const uint64_t* g2_tau_mont, int32_t* out_ok) {
    if (!ctx || !out_ok) return KZG_ERR_INVALID_ARG;
    if (n && (!commitments_xy_mont || !zs_mont || !ys_mont || !proofs_xy_mont || !r_powers_mont)) return KZG_ERR_INVALID_ARG;
    return verify_batch_core(ctx, commitments_xy_mont, zs_mont, ys_mont, proofs_xy_mont, r_powers_mont, n, g2_tau_mont, out_ok);
}

// ---- NTT ------------------------------------------------------------------------------------------
int32_t kzg_fr_ntt_device(kzg_ctx* ctx, void* d_data_mont, size_t n, int32_t inverse) {
    if (!ctx || (!d_data_mont && n)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc = ntt_run(ctx, d_data_mont, n, inverse != 0);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

int32_t kzg_fr_ntt(kzg_ctx* ctx, uint64_t* data_mont, size_t n, int32_t inverse) {
    if (!ctx || (!data_mont && n)) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * 32));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->poly[0].a.p, data_mont, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = ntt_run(ctx, ctx->poly[0].a.p, n, inverse != 0);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, hipMemcpyAsync(data_mont, ctx->poly[0].a.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

// ---- KZG surface ------------------------------------------------------------------------------------
int32_t kzg_commit_coeff_form(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* coeffs_mont, size_t n,
                              uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !srs || !out_xy_mont) return KZG_ERR_INVALID_ARG;
    if (n > srs->n) return KZG_ERR_POLY_LENGTH;                       // kzg.rs:112-116
    return msm_srs_common(ctx, srs, 0, coeffs_mont, false, n, out_xy_mont, out_is_infinity, nullptr);
}

int32_t kzg_commit_eval_form(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                             uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !srs || srs->ctx != ctx || !out_xy_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;             // kzg.rs:89-94
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO; // kzg.rs:265-269 (g1_ifft)
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    std::lock_guard<std::mutex> lk(ctx->mu);                            // (taken BEFORE the cache is looked at: kzg_srs_drop_lagrange frees the basis under this lock)
    if (const kzg_srs* cached = srs_cached_lagrange(srs, n))            // the reference's literal form: MSM over the Lagrange basis (kzg.rs:98-100);
        return msm_srs_locked(ctx, cached, 0, evals_mont, false, n, out_xy_mont, out_is_infinity, nullptr);   // large ones in two parts, the second upload hidden
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * 32));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->poly[0].a.p, evals_mont, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = ntt_run(ctx, ctx->poly[0].a.p, n, true);               // coefficients = IFFT(evaluations)
    if (rc != KZG_OK) return rc;
    return msm_run(ctx, srs_bases(srs, 0, n, ctx->msm_c_override == 0), ctx->poly[0].a.p, n, out_xy_mont, out_is_infinity, nullptr);
}

// ---- batched commitments: `count` polynomials of n coefficients / evaluations over one SRS in ONE kernel sequence -------------------
// The reference commits one blob per call (kzg.rs:84-125); a service that commits many small blobs pays the fixed depth of an MSM
// (~0.14-0.2 ms at 512..2048 coefficients: sort, ~13 dependent point additions, host epilogue) per blob.  Here the polynomials share the
// bases, so they are one MSM problem: width-8 NAF digits over the SRS's per-bit tables, 64 buckets per polynomial in one bucket
// array (msm.hip msm_run_batch_tables).  An SRS without per-bit tables (fewer than 2^15 points) gets them on the first batched call.
static int32_t commit_batch_device(kzg_ctx* ctx, kzg_srs* basis, const void* d_scalars, size_t n, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    uint4* bits = srs_bits(basis);
    if (!bits) {
        int32_t rc = srs_build_bit_tables(ctx, basis, true);
        if (rc != KZG_OK) return rc;
        bits = srs_bits(basis);
    }
    if (!bits) {                                  // no room for tables (or KZG_NO_NAF): one commitment after the other
        for (size_t k = 0; k < count; ++k) {
            int32_t rc = msm_run(ctx, srs_bases(basis, 0, n, ctx->msm_c_override == 0), static_cast<const char*>(d_scalars) + k * n * 32, n,
                                 out_xy + 8 * k, out_inf ? out_inf + k : nullptr, nullptr);
            if (rc != KZG_OK) return rc;
        }
        return KZG_OK;
    }
    MsmBases b;
    b.points = bits; b.table_stride = (uint32_t)basis->n; b.c = 7; b.W = 255; b.naf = true;
    return msm_run_batch_tables(ctx, b, d_scalars, n, count, out_xy, out_inf);
}
static int32_t commit_batch_common(kzg_ctx* ctx, kzg_srs* srs, const void* scalars, bool on_device, bool eval_form, size_t n, size_t count,
                                   uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx || !srs || srs->ctx != ctx || !out_xy || (n && count && !scalars)) return KZG_ERR_INVALID_ARG;
    if (n > srs->n) return eval_form ? KZG_ERR_SRS_CAPACITY_EXCEEDED : KZG_ERR_POLY_LENGTH;          // kzg.rs:89-94, :112-116
    if (eval_form && (n == 0 || (n & (n - 1)) != 0)) return KZG_ERR_NOT_POWER_OF_TWO;                // kzg.rs:265-269
    if (count == 0) return KZG_OK;
    if (n == 0) { memset(out_xy, 0, count * 64); if (out_inf) memset(out_inf, 1, count); return KZG_OK; }
    if (count > ((size_t)1 << 40) / n) return KZG_ERR_TOO_LARGE;
    kzg_srs* basis = srs;
    if (eval_form) {                                       // the reference's literal form: MSM over the Lagrange basis of n points, kept with the SRS
        int32_t rc = kzg_srs_cache_lagrange(ctx, srs, n);
        if (rc != KZG_OK) return rc;
        basis = srs_cached_lagrange(srs, n);
        if (!basis) return KZG_ERR_INVALID_ARG;             // dropped by another thread between the two calls
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const void* d = scalars;
    if (!on_device) {
        KZG_HIP_TRY(ctx, ctx->msm.scalars.reserve(count * n * 32 + 32));
        KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->msm.scalars.p, scalars, count * n * 32, hipMemcpyHostToDevice, ctx->stream));
        KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the launches run on the slots' streams
        d = ctx->msm.scalars.p;
    }
    return commit_batch_device(ctx, basis, d, n, count, out_xy, out_inf);
}
int32_t kzg_commit_coeff_form_batch(kzg_ctx* ctx, kzg_srs* srs, const uint64_t* coeffs_mont, size_t n, size_t count,
                                    uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    return commit_batch_common(ctx, srs, coeffs_mont, false, false, n, count, out_xy_mont, out_is_infinity);
}
int32_t kzg_commit_coeff_form_batch_device(kzg_ctx* ctx, kzg_srs* srs, const void* d_coeffs_mont, size_t n, size_t count,
                                           uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    return commit_batch_common(ctx, srs, d_coeffs_mont, true, false, n, count, out_xy_mont, out_is_infinity);
}
int32_t kzg_commit_eval_form_batch(kzg_ctx* ctx, kzg_srs* srs, const uint64_t* evals_mont, size_t n, size_t count,
                                   uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    return commit_batch_common(ctx, srs, evals_mont, false, true, n, count, out_xy_mont, out_is_infinity);
}
int32_t kzg_commit_eval_form_batch_device(kzg_ctx* ctx, kzg_srs* srs, const void* d_evals_mont, size_t n, size_t count,
                                          uint64_t* out_xy_mont, uint8_t* out_is_infinity) {
    return commit_batch_common(ctx, srs, d_evals_mont, true, true, n, count, out_xy_mont, out_is_infinity);
}

// ---- multi-GPU forms: this rank's SRS shard holds the powers [shard_lo, shard_lo + len(srs_shard)) -----------------------
int32_t kzg_commit_eval_form_partial(kzg_ctx* ctx, const kzg_srs* srs_shard, size_t shard_lo, const uint64_t* evals_mont, size_t n,
                                     uint64_t out_xyzz_mont[16]) {
    if (!ctx || !srs_shard || srs_shard->ctx->device != ctx->device || !out_xyzz_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * 32));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->poly[0].a.p, evals_mont, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = ntt_run(ctx, ctx->poly[0].a.p, n, true);               // every rank transforms the whole polynomial (32 B/element; not sharded)
    if (rc != KZG_OK) return rc;
    if (shard_lo >= n) { KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); memset(out_xyzz_mont, 0, 128); return KZG_OK; }
    const size_t len = std::min(srs_shard->n, n - shard_lo);
    return msm_run(ctx, srs_bases(srs_shard, 0, len, ctx->msm_c_override == 0), ctx->poly[0].a.as<uint4>() + 2 * shard_lo, len, nullptr, nullptr, out_xyzz_mont);
}

int32_t kzg_compute_proof_partial(kzg_ctx* ctx, const kzg_srs* srs_shard, size_t shard_lo, const uint64_t* evals_mont, size_t n,
                                  const uint64_t* roots_mont, size_t n_roots, const uint64_t z_mont[4],
                                  uint64_t out_xyzz_mont[16], uint64_t* out_y_mont) {
    (void)roots_mont;
    if (!ctx || !srs_shard || srs_shard->ctx != ctx || !out_xyzz_mont || !z_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return proof_run(ctx, srs_shard, evals_mont, n, z_mont, nullptr, nullptr, out_y_mont, true, shard_lo, out_xyzz_mont);
}

// ---- config 4 sharded by evaluation index over the Lagrange basis (lagrange.hip) --------------------------------------------------
int32_t kzg_commit_eval_form_lagrange_partial(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const uint64_t* evals_slice_mont, size_t len,
                                              uint64_t out_xyzz_mont[16]) {
    if (!ctx || !lagrange_shard || !out_xyzz_mont) return KZG_ERR_INVALID_ARG;
    if (len > lagrange_shard->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;              // kzg.rs:89-94
    return msm_srs_common(ctx, lagrange_shard, 0, evals_slice_mont, false, len, nullptr, nullptr, out_xyzz_mont);
}
int32_t kzg_commit_eval_form_lagrange_partial_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const void* d_evals_slice_mont, size_t len,
                                                     uint64_t out_xyzz_mont[16]) {
    if (!ctx || !lagrange_shard || !out_xyzz_mont) return KZG_ERR_INVALID_ARG;
    if (len > lagrange_shard->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    return msm_srs_common(ctx, lagrange_shard, 0, d_evals_slice_mont, true, len, nullptr, nullptr, out_xyzz_mont);
}
static int32_t lagrange_begin_common(kzg_ctx* ctx, const kzg_srs* shard, size_t lo, const void* evals, bool on_device, size_t len, size_t n,
                                     const uint64_t z[4], int32_t slot, int32_t commit_slot = -1) {
    if (!ctx || !shard || shard->ctx->device != ctx->device || !z || (len && !evals)) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;           // helpers.rs:485-487
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    if (lo > n || len > n - lo) return KZG_ERR_INVALID_ARG;
    if (len > shard->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                        // kzg.rs:89-94 (commit_eval_form of the quotient)
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_begin(ctx, shard, lo, evals, on_device, len, n, z, slot, commit_slot);
}
int32_t kzg_compute_proof_lagrange_begin(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont,
                                         size_t len, size_t n, const uint64_t z_mont[4], int32_t slot) {
    return lagrange_begin_common(ctx, lagrange_shard, shard_lo, evals_slice_mont, false, len, n, z_mont, slot);
}
int32_t kzg_compute_proof_lagrange_begin_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont,
                                                size_t len, size_t n, const uint64_t z_mont[4], int32_t slot) {
    return lagrange_begin_common(ctx, lagrange_shard, shard_lo, d_evals_slice_mont, true, len, n, z_mont, slot);
}
int32_t kzg_commit_and_prove_lagrange_begin(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont, size_t len,
                                            size_t n, const uint64_t z_mont[4], int32_t commit_slot, int32_t proof_slot) {
    if (commit_slot < 0) return KZG_ERR_INVALID_ARG;
    return lagrange_begin_common(ctx, lagrange_shard, shard_lo, evals_slice_mont, false, len, n, z_mont, proof_slot, commit_slot);
}
int32_t kzg_commit_and_prove_lagrange_begin_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont, size_t len,
                                                   size_t n, const uint64_t z_mont[4], int32_t commit_slot, int32_t proof_slot) {
    if (commit_slot < 0) return KZG_ERR_INVALID_ARG;
    return lagrange_begin_common(ctx, lagrange_shard, shard_lo, d_evals_slice_mont, true, len, n, z_mont, proof_slot, commit_slot);
}
int32_t kzg_compute_proof_lagrange_partial_y(kzg_ctx* ctx, int32_t slot, uint64_t out_ypart_mont[8]) {
    if (!ctx || !out_ypart_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_partial_y(ctx, slot, out_ypart_mont);
}
int32_t kzg_compute_proof_lagrange_continue(kzg_ctx* ctx, int32_t slot, const uint64_t y_mont[4]) {
    if (!ctx || !y_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_continue(ctx, slot, y_mont);
}
int32_t kzg_compute_proof_lagrange_end(kzg_ctx* ctx, int32_t slot, uint64_t out_part[32]) {
    if (!ctx || !out_part) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_end(ctx, slot, out_part, nullptr);
}
int32_t kzg_commit_and_prove_lagrange_end(kzg_ctx* ctx, int32_t slot, uint64_t out_commit_xyzz_mont[16], uint64_t out_part[32]) {
    if (!ctx || !out_part || !out_commit_xyzz_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_end(ctx, slot, out_part, out_commit_xyzz_mont);
}
int32_t kzg_compute_proof_lagrange_abort(kzg_ctx* ctx, int32_t slot) {
    if (!ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    lag_abort(ctx, slot);
    return KZG_OK;
}
int32_t kzg_compute_quotient_eval_on_domain(kzg_ctx* ctx, const uint64_t z_mont[4], const uint64_t* evals_mont, size_t n, const uint64_t value_mont[4],
                                            uint64_t out_quotient_mont[4]) {
    if (!ctx || !z_mont || !evals_mont || !value_mont || !out_quotient_mont) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lag_quotient_eval_on_domain(ctx, z_mont, evals_mont, n, value_mont, out_quotient_mont);
}
int32_t kzg_lagrange_fold_y(const uint64_t* yparts_mont, size_t count, size_t n, const uint64_t z_mont[4], uint64_t out_y_mont[4]) {
    if (!z_mont || !out_y_mont || (count && !yparts_mont)) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    return lag_fold_y(yparts_mont, count, n, z_mont, out_y_mont);
}
int32_t kzg_lagrange_fold_proof(const uint64_t* parts, size_t count, size_t n, const uint64_t z_mont[4], uint64_t out_xy_mont[8],
                                uint8_t* out_is_infinity) {
    if (!z_mont || !out_xy_mont || (count && !parts)) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    return lag_fold_proof(parts, count, n, z_mont, out_xy_mont, out_is_infinity);
}

static size_t next_pow2_sz(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

int32_t kzg_blob_to_fr(kzg_ctx* ctx, const uint8_t* blob_bytes, size_t len, uint64_t* out_mont, size_t cap, size_t* n_out) {
    if (!ctx || !n_out) return KZG_ERR_INVALID_ARG;
    const size_t n_elems = (len + 31) / 32;
    if (n_elems > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;                       // polynomial.rs:42-46
    const size_t n_padded = next_pow2_sz(n_elems);                                   // next_power_of_two(0) == 1
    *n_out = n_padded;                                                               // size query: out == NULL
    if (!out_mont || cap < n_padded || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    int32_t rc = blob_to_fr_run(ctx, blob_bytes, len, n_padded, &d);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, hipMemcpyAsync(out_mont, d, n_padded * 32, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

int32_t kzg_commit_blob(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len,
                        uint64_t out_xy_mont[8], uint8_t* out_is_infinity) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !out_xy_mont || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    const size_t n_elems = (len + 31) / 32;
    if (n_elems > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    const size_t n = next_pow2_sz(n_elems);
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                            // kzg.rs:89-94
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const kzg_srs* cached = srs_cached_lagrange(srs, n);
    {
        // over the cached Lagrange basis a large blob goes in TWO parts on two slots, as msm_srs_common does for scalars: bytes -> Fr and the MSM of the
        // first part run while this thread sits in the upload of the second (no IFFT ties the parts together)
        const bool idle = !ctx->slot_pending[0] && !ctx->slot_pending[1] && ctx->lag[0].phase == 0 && ctx->lag[1].phase == 0;
        if (cached && idle && n >= ((size_t)1 << 19) && n <= MSM_SPLIT_MAX && ctx->msm_c_override == 0 && srs_bits(cached)) {
            const size_t half = ((size_t)((double)n * (n >= ((size_t)1 << 20) ? 0.375 : 0.44)) + 255) / 256 * 256;
            const size_t e_lo[2] = {0, half}, e_n[2] = {half, n - half};
            int32_t rc = KZG_OK;
            int begun = 0;
            for (int h = 0; h < 2 && rc == KZG_OK; ++h) {
                hipStream_t st = nullptr;
                rc = msm_slot_stream(ctx, h, &st);
                if (rc != KZG_OK) break;
                MsmWorkspace& ws = ctx->slot_msm(h);
                const size_t b_lo = std::min(len, e_lo[h] * 32), b_hi = std::min(len, (e_lo[h] + e_n[h]) * 32);
                void* d_part = nullptr;
                rc = blob_to_fr_run(ctx, blob_bytes + b_lo, b_hi - b_lo, e_n[h], &d_part, st, &ws.blob, &ws.scalars);
                if (rc == KZG_OK) rc = msm_begin(ctx, h, srs_bases(cached, e_lo[h], e_n[h], true), d_part, e_n[h]);
                if (rc == KZG_OK) ++begun;
            }
            kzg_host::Xyzz parts[2] = {kzg_host::xyzz_inf(), kzg_host::xyzz_inf()};
            for (int h = 0; h < begun; ++h) {
                uint64_t w[16];
                const int32_t r2 = msm_end(ctx, h, nullptr, nullptr, w);
                if (r2 == KZG_OK) memcpy(&parts[h], w, 128); else if (rc == KZG_OK) rc = r2;
            }
            if (rc != KZG_OK) return rc;
            kzg_host::xyzz_to_affine(kzg_host::xyzz_add(parts[0], parts[1]), out_xy_mont, out_is_infinity);
            return KZG_OK;
        }
    }
    void* d = nullptr;
    int32_t rc = blob_to_fr_run(ctx, blob_bytes, len, n, &d);                        // Blob::to_polynomial_eval_form
    if (rc != KZG_OK) return rc;
    if (cached)                                                                      // commit_eval_form's literal form (kzg.rs:98-100): MSM over the cached Lagrange basis
        return msm_run(ctx, srs_bases(cached, 0, n, ctx->msm_c_override == 0), d, n, out_xy_mont, out_is_infinity, nullptr);
    rc = ntt_run(ctx, d, n, true);                                                   // commit_eval_form: IFFT ...
    if (rc != KZG_OK) return rc;
    return msm_run(ctx, srs_bases(srs, 0, n, ctx->msm_c_override == 0), d, n, out_xy_mont, out_is_infinity, nullptr);   // ... + MSM
}

// asynchronous forms of commit_eval_form / commit_blob: the whole chain (H2D, bytes -> Fr, IFFT, MSM) goes onto the slot's stream
static int32_t commit_begin_common(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, const uint8_t* blob_bytes, size_t len,
                                   size_t n, int32_t slot) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS) return KZG_ERR_INVALID_ARG;
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                            // kzg.rs:89-94
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 24)) return KZG_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    MsmWorkspace& ws = ctx->slot_msm(slot);
    void* d = nullptr;
    if (blob_bytes || !evals_mont) {
        rc = blob_to_fr_run(ctx, blob_bytes, len, n, &d, st, &ws.blob, &ws.scalars);
        if (rc != KZG_OK) return rc;
    } else {
        KZG_HIP_TRY(ctx, ws.scalars.reserve(n * 32 + 32));
        KZG_HIP_TRY(ctx, hipMemcpyAsync(ws.scalars.p, evals_mont, n * 32, hipMemcpyHostToDevice, st));
        d = ws.scalars.p;
    }
    if (const kzg_srs* cached = srs_cached_lagrange(srs, n))                         // no IFFT: MSM over the cached Lagrange basis
        return msm_begin(ctx, slot, srs_bases(cached, 0, n, ctx->msm_c_override == 0), d, n);
    NttTables tb;                                                                    // make sure the tables exist before the slot stream reads them
    int log_n = 0; while (((size_t)1 << log_n) < n) ++log_n;
    if (n > 1) { rc = ntt_get_tables(ctx, log_n, true, &tb); if (rc != KZG_OK) return rc; }
    rc = ntt_run(ctx, d, n, true, st, &ctx->slot_ntt(slot));
    if (rc != KZG_OK) return rc;
    return msm_begin(ctx, slot, srs_bases(srs, 0, n, ctx->msm_c_override == 0), d, n);
}
int32_t kzg_commit_eval_form_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n, int32_t slot) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || !evals_mont) return KZG_ERR_INVALID_ARG;
    return commit_begin_common(ctx, srs, evals_mont, nullptr, 0, n, slot);
}
int32_t kzg_commit_blob_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, int32_t slot) {
    if (!ctx || !srs || srs->ctx->device != ctx->device || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    const size_t n_elems = (len + 31) / 32;
    if (n_elems > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    return commit_begin_common(ctx, srs, nullptr, blob_bytes, len, next_pow2_sz(n_elems), slot);
}

int32_t kzg_g1_ifft(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint64_t* out_xy_mont) {
    if (!ctx || !srs || srs->ctx != ctx) return KZG_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;             // kzg.rs:265-269
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;                               // kzg.rs:275-278
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    if (!out_xy_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return g1_ifft_run(ctx, srs, n, out_xy_mont);
}

int32_t kzg_compute_proof(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                          const uint64_t* roots_mont, size_t n_roots, const uint64_t z_mont[4],
                          uint64_t out_xy_mont[8], uint8_t* out_is_infinity, uint64_t* out_y_mont) {
    (void)roots_mont;   // KZG::expanded_roots_of_unity can only hold the canonical domain (kzg.rs:65-72): its length is what is checked
    if (!ctx || !srs || srs->ctx != ctx || !out_xy_mont || !z_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;                    // kzg.rs:135-139, :222-226
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return proof_run(ctx, srs, evals_mont, n, z_mont, out_xy_mont, out_is_infinity, out_y_mont, true, 0, nullptr);
}

int32_t kzg_compute_proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                                const uint64_t* roots_mont, size_t n_roots, const uint64_t z_mont[4], int32_t slot) {
    (void)roots_mont;
    if (!ctx || !srs || srs->ctx != ctx || !z_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;
    if (n > ((size_t)1 << 24)) return KZG_ERR_TOO_LARGE;
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return proof_begin(ctx, srs, evals_mont, n, z_mont, slot);
}
int32_t kzg_compute_proof_end(kzg_ctx* ctx, int32_t slot, uint64_t out_xy_mont[8], uint8_t* out_is_infinity, uint64_t* out_y_mont) {
    if (!ctx || !out_xy_mont) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return proof_end(ctx, slot, out_xy_mont, out_is_infinity, out_y_mont);
}

// ---- Fiat-Shamir challenge and blob proofs (helpers.rs:411-472, kzg.rs:288-309) --------------------------------------------
namespace {
const uint64_t FR_R2_WORDS[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};   // 2^512 mod r
}  // namespace
}  // extern "C"
namespace kzg {   // (shared with blobstream.hip)

// Absorbs  tag || u64be(n) || n x 32 bytes, the evaluations of Blob::to_polynomial_eval_form in to_byte_array form: every
// 32-byte big-endian chunk of the blob reduced mod r (helpers.rs:40-57 -> :80-119), zero elements up to the next power of two.
// Canonical chunks (the normal case) are hashed straight from the caller's buffer.
void challenge_absorb_prefix(kzg_host::Sha256& sh, const uint8_t* blob, size_t len, size_t n_padded) {
    kzg_host::TranscriptPrefix gen(blob, len, n_padded);                             // host_transcript.h
    kzg_host::sha256_absorb(sh, gen);
}
// ark-serialize compressed G1Affine (helpers.rs:456-459): x little-endian, bit 7 of the last byte = y is the larger root, bit 6 = infinity
void g1_serialize_compressed_ark(const kzg_host::G1& p, uint8_t out[32]) {
    using namespace kzg_host;
    memset(out, 0, 32);
    if (p.inf) { out[31] = 0x40; return; }
    Fq one_plain = {{1, 0, 0, 0}};
    Fq x = mul(p.x, one_plain), y = mul(p.y, one_plain);                             // Montgomery -> canonical
    memcpy(out, x.l, 32);                                                            // little-endian host
    Fq ny = sub(FQ_P, y);                                                            // -y (y != 0 on this curve)
    bool larger = false;
    for (int i = 3; i >= 0; --i) if (y.l[i] != ny.l[i]) { larger = y.l[i] > ny.l[i]; break; }
    if (larger) out[31] |= 0x80;
}
void challenge_finish(kzg_host::Sha256& sh, const kzg_host::G1& commitment, uint64_t out_z_mont[4]) {
    using namespace kzg_host;
    uint8_t cb[32], dig[32];
    g1_serialize_compressed_ark(commitment, cb);
    sha256_update(sh, cb, 32);
    sha256_final(sh, dig);
    uint64_t w[4];
    for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int b = 0; b < 8; ++b) v = (v << 8) | dig[8 * (3 - i) + b]; w[i] = v; }
    while (fr_geq_r(w)) fr_sub_r(w);                                                // Fr::from_be_bytes_mod_order (helpers.rs:382-390)
    fr_mul(w, FR_R2_WORDS, out_z_mont);                                             // canonical -> Montgomery
}
size_t blob_padded_len(size_t len) { size_t e = (len + 31) / 32, p = 1; while (p < e) p <<= 1; return p; }

}  // namespace kzg
extern "C" {

int32_t kzg_validate_g1_point(const uint64_t xy_mont[8]) {
    if (!xy_mont) return KZG_ERR_INVALID_ARG;
    return kzg_host::g1_on_curve(kzg_host::g1_from_wire(xy_mont)) ? KZG_OK : KZG_ERR_G1_NOT_ON_CURVE;
}

int32_t kzg_hash_to_field_element(const uint8_t* msg, size_t len, uint64_t out_mont[4]) {
    if (!out_mont || (len && !msg)) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    Sha256 sh;
    sha256_init(sh);
    if (len) sha256_update(sh, msg, len);
    uint8_t dig[32];
    sha256_final(sh, dig);
    uint64_t w[4];
    for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int b = 0; b < 8; ++b) v = (v << 8) | dig[8 * (3 - i) + b]; w[i] = v; }
    while (fr_geq_r(w)) fr_sub_r(w);
    fr_mul(w, FR_R2_WORDS, out_mont);
    return KZG_OK;
}

int32_t kzg_compute_challenge(const uint8_t* blob_bytes, size_t len, const uint64_t commitment_xy_mont[8], uint64_t out_z_mont[4]) {
    if (!commitment_xy_mont || !out_z_mont || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    if ((len + 31) / 32 > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;
    kzg_host::G1 c = kzg_host::g1_from_wire(commitment_xy_mont);
    if (!kzg_host::g1_on_curve(c)) return KZG_ERR_G1_NOT_ON_CURVE;                   // validate_g1_point (helpers.rs:694-708; cofactor 1)
    kzg_host::Sha256 sh;
    kzg_host::sha256_init(sh);
    challenge_absorb_prefix(sh, blob_bytes, len, blob_padded_len(len));
    challenge_finish(sh, c, out_z_mont);
    return KZG_OK;
}

// shared body: `commitment_in` given (compute_blob_proof) or computed here beside the transcript hash (commit + proof)
static int32_t blob_proof_common(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                                 const uint64_t* commitment_in, uint64_t* out_commitment_xy, uint8_t* out_commitment_inf,
                                 uint64_t out_proof_xy[8], uint8_t* out_proof_inf, uint64_t* out_z_mont, uint64_t* out_y_mont) {
    if (!ctx || !srs || srs->ctx != ctx || !out_proof_xy || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    const size_t n_elems = (len + 31) / 32;
    if (n_elems > ((size_t)1 << 28)) return KZG_ERR_TOO_LARGE;                       // polynomial.rs:42-46
    const size_t n = blob_padded_len(len);
    kzg_host::G1 commitment;
    if (commitment_in) {
        commitment = kzg_host::g1_from_wire(commitment_in);
        if (!kzg_host::g1_on_curve(commitment)) return KZG_ERR_G1_NOT_ON_CURVE;      // kzg.rs:295
    }
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;                                   // kzg.rs:135-139 (compute_proof_impl)
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                            // kzg.rs:89-94 (commit_eval_form of the quotient)
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->slot_pending[0]) { ctx->last_error = "a kzg_*_begin on slot 0 is still in flight: call its end first"; return KZG_ERR_INVALID_ARG; }
    // the transcript prefix does not depend on the commitment: hash it on a host thread while the GPU works
    kzg_host::Sha256 sh;
    kzg_host::sha256_init(sh);
    std::thread hasher([&] { challenge_absorb_prefix(sh, blob_bytes, len, n); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{hasher};
    PolySet& set = ctx->poly[0];
    void* d_evals = nullptr;
    int32_t rc = blob_to_fr_run(ctx, blob_bytes, len, n, &d_evals, ctx->stream, &set.c, &set.a);    // evaluations (wire) in set.a
    if (rc != KZG_OK) return rc;
    if (!commitment_in) {
        // commit_blob on a copy of the evaluations (the IFFT is in place); runs beside the hash
        uint64_t cxy[8]; uint8_t cinf = 0;
        if (const kzg_srs* cached = srs_cached_lagrange(srs, n)) {                   // the evaluations themselves are the scalars: no copy, no IFFT
            rc = msm_run(ctx, srs_bases(cached, 0, n, ctx->msm_c_override == 0), d_evals, n, cxy, &cinf, nullptr);
        } else {
            KZG_HIP_TRY(ctx, ctx->msm.scalars.reserve(n * 32 + 32));
            KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->msm.scalars.p, d_evals, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
            rc = ntt_run(ctx, ctx->msm.scalars.p, n, true);
            if (rc != KZG_OK) return rc;
            rc = msm_run(ctx, srs_bases(srs, 0, n, ctx->msm_c_override == 0), ctx->msm.scalars.p, n, cxy, &cinf, nullptr);
        }
        if (rc != KZG_OK) return rc;
        commitment = kzg_host::g1_from_wire(cxy);
        if (out_commitment_xy) memcpy(out_commitment_xy, cxy, 64);
        if (out_commitment_inf) *out_commitment_inf = cinf;
    }
    hasher.join();
    uint64_t z[4];
    challenge_finish(sh, commitment, z);
    if (out_z_mont) memcpy(out_z_mont, z, 32);
    return proof_run(ctx, srs, nullptr, n, z, out_proof_xy, out_proof_inf, out_y_mont, true, 0, nullptr);   // evaluations already in set.a
}

int32_t kzg_compute_blob_proof(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                               const uint64_t commitment_xy_mont[8], uint64_t out_proof_xy_mont[8], uint8_t* out_is_infinity,
                               uint64_t* out_z_mont, uint64_t* out_y_mont) {
    if (!commitment_xy_mont) return KZG_ERR_INVALID_ARG;
    return blob_proof_common(ctx, srs, blob_bytes, len, n_roots, commitment_xy_mont, nullptr, nullptr, out_proof_xy_mont, out_is_infinity,
                             out_z_mont, out_y_mont);
}

int32_t kzg_commit_and_prove_blob(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                                  uint64_t out_commitment_xy_mont[8], uint8_t* out_commitment_is_infinity,
                                  uint64_t out_proof_xy_mont[8], uint8_t* out_proof_is_infinity, uint64_t* out_z_mont, uint64_t* out_y_mont) {
    if (!out_commitment_xy_mont) return KZG_ERR_INVALID_ARG;
    return blob_proof_common(ctx, srs, blob_bytes, len, n_roots, nullptr, out_commitment_xy_mont, out_commitment_is_infinity,
                             out_proof_xy_mont, out_proof_is_infinity, out_z_mont, out_y_mont);
}

int32_t kzg_evaluate_polynomial_in_evaluation_form(kzg_ctx* ctx, const uint64_t* evals_mont, size_t n,
                                                   const uint64_t z_mont[4], uint64_t out_y_mont[4]) {
    if (!ctx || !z_mont || !out_y_mont || (n && !evals_mont)) return KZG_ERR_INVALID_ARG;
    if (n == 0) return KZG_ERR_ZERO_LENGTH;
    if ((n & (n - 1)) != 0) return KZG_ERR_INVALID_INPUT_LENGTH;      // helpers.rs:485-487
    if (n > ((size_t)1 << 28)) return KZG_ERR_SRS_LENGTH;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return proof_run(ctx, nullptr, evals_mont, n, z_mont, nullptr, nullptr, out_y_mont, false, 0, nullptr);
}

int32_t kzg_calculate_roots_of_unity(kzg_ctx* ctx, uint64_t length_of_data_after_padding, uint64_t* out_mont, size_t cap, size_t* n_out) {
    if (!ctx || !n_out) return KZG_ERR_INVALID_ARG;
    if (length_of_data_after_padding == 0) return KZG_ERR_ZERO_LENGTH;                  // helpers.rs:554-558
    uint64_t elems = (length_of_data_after_padding + 31) / 32;
    if (elems > ((uint64_t)1 << 28)) return KZG_ERR_SRS_LENGTH;                          // helpers.rs:560-566
    size_t n = 1;
    while (n < elems) n <<= 1;
    *n_out = n;
    if (!out_mont || cap < n) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return roots_run(ctx, out_mont, n);
}

// ---- batch verification end to end (verifier/src/batch.rs:16-69, :76-168; primitives/src/helpers.rs:613-662) ---------------------
namespace {

const char RC_BATCH_DOMAIN[] = "EIGENDA_RCKZGBATCH___V1_";     // primitives/src/consts.rs:11 (24 bytes)
constexpr size_t VB_GROUP_BYTES = (size_t)256 << 20;           // packed blob bytes per GPU round
constexpr uint32_t VB_BATCH_MAX_LOG = 12;                      // = VB_MAX_LOG of poly.hip: larger blobs take the single-polynomial path
struct VbMeta { uint64_t off; uint32_t len; uint32_t log_n; }; // = VbBlob of poly.hip

void fr_wire_to_be_bytes(const uint64_t wire[4], uint8_t out[32]) {
    uint64_t k[4];
    kzg_host::fr_wire_to_canonical(wire, k);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(k[3 - i] >> (8 * (7 - b)));
}
void digest_to_fr_wire(const uint8_t dig[32], uint64_t out[4]) {            // hash_to_field_element (helpers.rs:382-390)
    uint64_t w[4];
    for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int b = 0; b < 8; ++b) v = (v << 8) | dig[8 * (3 - i) + b]; w[i] = v; }
    while (kzg_host::fr_geq_r(w)) kzg_host::fr_sub_r(w);
    kzg_host::fr_mul(w, FR_R2_WORDS, out);
}
// Upper bound of the pool, whatever the core count (KZG_HOST_THREADS_MAX).  Measured on the 256-thread host of an MI355X box, 4 096-row batch verification
// end to end (tools/trace_batch_verify.py): 32 threads 6.3-6.5 ms, 48 5.4-5.5, 64 5.3-5.6, 96 5.8-5.9, 128 6.0-6.5 (the transcripts scale, the upload and the
// serial parts do not, and past 64 the pool's wake-ups cost more than they gain) -> 48.
// CPUs this process may use on average: the CFS bandwidth quota of its cgroup (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), 0 = none.
// A pool wider than the quota allows runs fine for one call, but back-to-back calls then spend the period's budget early and the kernel
// freezes EVERY thread of the cgroup until the next 100 ms period: that was the 20-34 ms tail of batch verification (1 call in 9; DESIGN.md
// section 6, profiles/r06_batch_verify_tail.md: every slow call coincides with a throttled period in cpu.stat, none without).
double cgroup_cpu_quota() {
    static const double q = []() -> double {
        auto read2 = [](const char* path, char a[64], char b[64]) {
            FILE* f = fopen(path, "r");
            if (!f) return 0;
            const int got = fscanf(f, "%63s %63s", a, b);
            fclose(f);
            return got;
        };
        char a[64] = {0}, b[64] = {0};
        if (read2("/sys/fs/cgroup/cpu.max", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) return atof(a) / atof(b);
        char c[64] = {0}, d[64] = {0};
        if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", a, b) >= 1 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", c, d) >= 1 && atof(a) > 0 && atof(c) > 0)
            return atof(a) / atof(c);
        return 0.0;
    }();
    return q;
}
unsigned host_threads_cap() {
    // 48 threads at most; under a CPU quota of Q, Q threads: the pool then never draws more than the quota, whatever else the process runs (HIP's
    // helper threads, the caller's own).  Measured on a 16-CPU quota, 4 096 blobs per call, 60 calls each: 48 threads median 6.2 ms / max 24 / 123 ms of CPU per
    // call; 32: 6.1 / 15 / 109; 24: 6.8 / 7.2 / 100; 20: 6.9 / 7.0 / 88 (but 13.7 once in 40 calls inside bench.py); 16: 7.7 / 7.8 / 84; 12: 9.4 / 9.8 / 86.
    // KZG_HOST_THREADS_MAX overrides.
    static const unsigned cap = []() {
        if (opts().host_threads_max >= 1 && opts().host_threads_max <= 256) return (unsigned)opts().host_threads_max;
        unsigned c = 48;
        const double q = cgroup_cpu_quota();
        if (q > 0) c = std::min<unsigned>(c, std::max<unsigned>(2u, (unsigned)q));
        return c;
    }();
    return cap;
}
unsigned host_threads(size_t jobs) {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 4;
    if (t > host_threads_cap()) t = host_threads_cap();
    if (opts().host_threads > 0) t = (unsigned)opts().host_threads;                                         // exactly that many (measurements)
    if ((size_t)t > jobs) t = (unsigned)jobs;
    return t ? t : 1;
}
// run job(i) for i in [0, n) on a pool of host threads (the calling thread included)
// A persistent pool of host threads (created on first use, parked on a condition variable between calls): spawning 31 threads
// per call cost 0.3-1 ms, three times per batch verification.  run(n, job) executes job(i) for i in [0, n) on the pool AND the
// calling thread; one run at a time.
class HostPool {
public:
    static HostPool& get() { static HostPool pool; return pool; }
    void run(size_t n, const std::function<void(size_t)>& job) {
        if (n == 0) return;
        const unsigned T = host_threads(n);
        if (T <= 1) { for (size_t i = 0; i < n; ++i) job(i); return; }
        std::lock_guard<std::mutex> one(run_mu_);
        ensure(T - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job; n_ = n; next_.store(0, std::memory_order_relaxed);
            wanted_ = T - 1; joined_ = 0; running_ = 0;
            ++generation_;
        }
        cv_work_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        wanted_ = 0;                                               // late wakers of this generation find nothing to join
        cv_done_.wait(lk, [&] { return running_ == 0; });
        job_ = nullptr;
    }
private:
    HostPool() = default;
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_work_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void ensure(unsigned count) {
        while (threads_.size() < count)
            threads_.emplace_back([this] { pthread_setname_np(pthread_self(), "kzg-pool"); loop(); });     // visible in /proc/<pid>/task/*/comm (tools/trace_batch_verify.py)
    }
    void work() { for (;;) { const size_t i = next_.fetch_add(1, std::memory_order_relaxed); if (i >= n_) return; (*job_)(i); } }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (joined_ >= wanted_) continue;                      // this run wants fewer threads than the pool has
            ++joined_; ++running_;
            lk.unlock();
            work();
            lk.lock();
            if (--running_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    unsigned wanted_ = 0, joined_ = 0, running_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};
void parallel_for(size_t n, const std::function<void(size_t)>& job) { HostPool::get().run(n, job); }
}  // namespace
}  // extern "C"
namespace kzg { void host_parallel_for(size_t n, const std::function<void(size_t)>& job) { parallel_for(n, job); } }
extern "C" {
namespace {

// The data-parallel front end of verify_blob_kzg_proof_batch: z_i = compute_challenge(blob_i, C_i), y_i = p_i(z_i) for all n blobs.
// Transcripts: n independent SHA-256 streams on a pool of host threads (each also packs its blob into the pinned staging buffer);
// evaluations: two GPU launches for all blobs of up to 4096 elements (poly.hip k_vb_prep / k_vb_eval), the single-polynomial path
// for the rest.  `validated` = the caller has already checked every commitment (batch.rs:29-37); otherwise compute_challenge's own
// validate_g1_point (helpers.rs:413) is reported in blob order.
// commitments == nullptr: zs are INPUTS (kzg_evaluate_blobs_in_evaluation_form_batch), nothing is hashed.
int32_t challenges_and_evaluations(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* lens, const uint64_t* commitments, size_t n,
                                   bool validated, uint64_t* zs, uint64_t* ys) {
    using namespace kzg_host;
    // per-blob guards in the reference's order (helpers.rs:634-645): to_polynomial_eval_form (TOO_LARGE), compute_challenge
    // (validate_g1_point), evaluate_polynomial_in_evaluation_form -> calculate_roots_of_unity (ZERO_LENGTH)
    std::vector<int32_t> status(n, KZG_OK);
    std::vector<VbMeta> meta(n);
    std::vector<size_t> group_end;                            // blob index where each GPU round ends
    // (KZG_VB_GROUP_BYTES / KZG_VB_CHUNK_BYTES: staging granularity; tests shrink them so that small batches take the multi-round / multi-chunk paths)
    const size_t group_bytes = opts().vb_group_bytes ? opts().vb_group_bytes : VB_GROUP_BYTES;
    const size_t chunk_bytes = opts().vb_chunk_bytes ? opts().vb_chunk_bytes : ((size_t)16 << 20);
    size_t off = 0;
    for (size_t i = 0; i < n; ++i) {
        const size_t elems = (lens[i] + 31) / 32;
        if (lens[i] && !blobs[i]) return KZG_ERR_INVALID_ARG;
        if (elems > ((size_t)1 << 28)) { status[i] = KZG_ERR_TOO_LARGE; meta[i] = VbMeta{0, 0, 99}; continue; }
        size_t np = 1; uint32_t lg = 0;
        while (np < elems) { np <<= 1; ++lg; }
        const bool batched = lens[i] != 0 && lg <= VB_BATCH_MAX_LOG;
        const size_t span = batched ? elems * 32 : 0;
        if (off && off + span > group_bytes) { group_end.push_back(i); off = 0; }
        meta[i] = VbMeta{(uint64_t)off, (uint32_t)(batched ? lens[i] : 0), batched ? lg : 99u};
        off += span;
    }
    group_end.push_back(n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->slot_pending[0]) { ctx->last_error = "a kzg_*_begin on slot 0 is still in flight: call its end first"; return KZG_ERR_INVALID_ARG; }
    std::vector<uint8_t> fallback(n, 0);
    size_t g0 = 0;
    for (size_t g1 : group_end) {
        size_t bytes = 0;
        for (size_t i = g0; i < g1; ++i) if (meta[i].log_n != 99u) bytes = std::max(bytes, (size_t)meta[i].off + ((size_t)meta[i].len + 31) / 32 * 32);
        const size_t bytes_al = (bytes + 4095) / 4096 * 4096, small_bytes = (g1 - g0) * 48 + 4096;   // packed blobs | challenges + descriptors
        if (bytes_al + small_bytes > ctx->vb_pinned_bytes) {
            if (ctx->vb_pinned) { (void)hipHostFree(ctx->vb_pinned); ctx->vb_pinned = nullptr; ctx->vb_pinned_bytes = 0; }
            const size_t cap = bytes_al + bytes_al / 4 + 2 * small_bytes;
            KZG_HIP_TRY(ctx, hipHostMalloc(&ctx->vb_pinned, cap, hipHostMallocDefault));
            ctx->vb_pinned_bytes = cap;
        }
        uint8_t* stage = static_cast<uint8_t*>(ctx->vb_pinned);
        RoctxPhases phases;
        phases.begin("kzg:batch_verify:transcripts + pack (host threads) | uploads + evaluation kernels (GPU)");
        const auto t_hash0 = std::chrono::steady_clock::now();
        const size_t nb = g1 - g0;
        int32_t rc = vb_evaluate_setup(ctx, bytes, nb);
        if (rc != KZG_OK) return rc;
        // Chunks of ~16 MiB of packed bytes: the thread that finishes the LAST blob of a chunk enqueues the chunk's upload and its two
        // kernels, so the PCIe transfer and the evaluations run beside the hashing of the later blobs (3 ms of host work and 3 ms of
        // upload + kernels one after the other before).  The pool hands blobs out in index order, so chunks complete roughly in order.
        std::vector<size_t> chunk_of(nb), chunk_lo, chunk_hi;
        {
            size_t acc_bytes = 0;
            for (size_t k = 0; k < nb; ++k) {
                if (chunk_lo.empty() || acc_bytes >= chunk_bytes) { chunk_lo.push_back(k); chunk_hi.push_back(k); acc_bytes = 0; }
                chunk_of[k] = chunk_lo.size() - 1;
                chunk_hi.back() = k + 1;
                if (meta[g0 + k].log_n != 99u) acc_bytes += ((size_t)meta[g0 + k].len + 31) / 32 * 32;
            }
        }
        std::vector<std::atomic<uint32_t>> left(chunk_lo.size());
        for (size_t c = 0; c < chunk_lo.size(); ++c) left[c].store((uint32_t)(chunk_hi[c] - chunk_lo[c]));
        std::atomic<int32_t> enqueue_rc{KZG_OK};
        const bool trace_chunks = opts().vb_trace >= 2;
        std::vector<double> enq_at(chunk_lo.size(), 0.0), enq_took(chunk_lo.size(), 0.0);
        std::vector<hipEvent_t> chunk_ev(trace_chunks ? chunk_lo.size() : 0);
        hipEvent_t ev0 = nullptr;
        if (trace_chunks) { for (auto& e : chunk_ev) (void)hipEventCreate(&e); (void)hipEventCreate(&ev0); (void)hipEventRecord(ev0, ctx->stream); }
        // Pool jobs: PAIRS of blobs of one upload chunk with similar transcript lengths (sorted inside the chunk), so that the two SHA-256 streams of a
        // job run interleaved in one thread to the end (host_transcript.h: 1.5 x the single-stream rate); chunks stay in index order, so they still
        // complete -- and go up -- roughly in order.  A blob that is not hashed (bad status, no commitments) rides along as a job of its own.
        std::vector<std::pair<uint32_t, uint32_t>> jobs;                 // blob indices relative to g0; second = UINT32_MAX: a single
        jobs.reserve(nb / 2 + chunk_lo.size() + 1);
        {
            std::vector<uint32_t> order;
            for (size_t c = 0; c < chunk_lo.size(); ++c) {
                order.clear();
                for (size_t k = chunk_lo[c]; k < chunk_hi[c]; ++k) order.push_back((uint32_t)k);
                std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                    const size_t lx = lens[g0 + x], ly = lens[g0 + y];
                    return lx != ly ? lx > ly : x < y;
                });
                for (size_t q = 0; q < order.size(); q += 2)
                    jobs.emplace_back(order[q], q + 1 < order.size() && commitments ? order[q + 1] : UINT32_MAX);
                if (!commitments) for (size_t q = 1; q < order.size(); q += 2) jobs.emplace_back(order[q], UINT32_MAX);
            }
        }
        parallel_for(jobs.size(), [&](size_t jdx) {
            auto done = [&](size_t k) {
                const size_t c = chunk_of[k];
                if (left[c].fetch_sub(1, std::memory_order_acq_rel) == 1) {
                    const auto e0 = std::chrono::steady_clock::now();
                    const int32_t r = vb_evaluate_enqueue(ctx, stage, meta.data() + g0, nb, chunk_lo[c], chunk_hi[c], zs + 4 * g0, stage + bytes_al);
                    if (r != KZG_OK) enqueue_rc.store(r);
                    if (trace_chunks) {
                        (void)hipEventRecord(chunk_ev[c], ctx->stream);
                        enq_at[c] = std::chrono::duration<double, std::milli>(e0 - t_hash0).count();
                        enq_took[c] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e0).count();
                    }
                }
            };
            // per blob: guards in the reference's order, packing; returns the bytes to hash (nullptr: nothing to hash for this blob)
            auto prepare = [&](size_t k, G1& c) -> const uint8_t* {
                const size_t i = g0 + k;
                if (status[i] != KZG_OK) return nullptr;
                if (commitments) {
                    c = g1_from_wire(commitments + 8 * i);
                    if (!validated && !g1_on_curve(c)) { status[i] = KZG_ERR_G1_NOT_ON_CURVE; return nullptr; }
                }
                if (lens[i] == 0) { status[i] = KZG_ERR_ZERO_LENGTH; return nullptr; }
                const uint8_t* src = blobs[i];
                if (meta[i].log_n != 99u) {                       // pack (zero-filled to the 32-byte chunk) and hash the packed copy
                    uint8_t* dst = stage + meta[i].off;
                    const size_t span = (lens[i] + 31) / 32 * 32;
                    memcpy(dst, blobs[i], lens[i]);                   // (a probe build without this copy: 3.6-4.6 ms for the phase against 4.0-4.4 with it -- inside the noise)
                    if (span > lens[i]) memset(dst + lens[i], 0, span - lens[i]);
                    src = dst;
                }
                return commitments ? src : nullptr;
            };
            const size_t ka = jobs[jdx].first, kb = jobs[jdx].second;
            G1 ca, cb;
            const uint8_t* pa = prepare(ka, ca);
            const uint8_t* pb = kb != UINT32_MAX ? prepare(kb, cb) : nullptr;
            if (pa && pb) {
                Sha256 sa, sb;
                sha256_init(sa); sha256_init(sb);
                TranscriptPrefix ga(pa, lens[g0 + ka], blob_padded_len(lens[g0 + ka])), gb(pb, lens[g0 + kb], blob_padded_len(lens[g0 + kb]));
                sha256_absorb_x2(sa, ga, sb, gb);
                challenge_finish(sa, ca, zs + 4 * (g0 + ka));
                challenge_finish(sb, cb, zs + 4 * (g0 + kb));
            } else {
                for (int w = 0; w < 2; ++w) {
                    const uint8_t* p1 = w ? pb : pa;
                    if (!p1) continue;
                    const size_t k1 = w ? kb : ka;
                    Sha256 sh;
                    sha256_init(sh);
                    challenge_absorb_prefix(sh, p1, lens[g0 + k1], blob_padded_len(lens[g0 + k1]));
                    challenge_finish(sh, w ? cb : ca, zs + 4 * (g0 + k1));
                }
            }
            done(ka);
            if (kb != UINT32_MAX) done(kb);
        });
        const auto t_hash_done = std::chrono::steady_clock::now();
        rc = vb_evaluate_finish(ctx, nb, ys + 4 * g0, fallback.data() + g0);         // (also drains the stream before any early return below)
        if (trace_chunks) {
            for (size_t c = 0; c < chunk_lo.size(); ++c) {
                float gpu_ms = 0;
                (void)hipEventElapsedTime(&gpu_ms, ev0, chunk_ev[c]);
                fprintf(stderr, "    chunk %zu blobs [%zu, %zu): enqueued at %.3f ms (call took %.3f ms), done on the GPU %.3f ms after the first enqueue point\n", c, chunk_lo[c],
                        chunk_hi[c], enq_at[c], enq_took[c], gpu_ms);
                (void)hipEventDestroy(chunk_ev[c]);
            }
            (void)hipEventDestroy(ev0);
        }
        if (enqueue_rc.load() != KZG_OK) return enqueue_rc.load();
        if (rc != KZG_OK) return rc;
        for (size_t i = g0; i < g1; ++i) if (status[i] != KZG_OK) return status[i];      // the first failing blob, in order
        {
            const bool trace = opts().vb_trace != 0;
            if (trace)
                fprintf(stderr, "  blobs [%zu, %zu): %zu packed bytes in %zu chunks; transcripts + pack %.3f ms (%u host threads, uploads and kernels beside them), "
                        "the rest of the GPU work + D2H %.3f ms\n", g0, g1, bytes, chunk_lo.size(),
                        std::chrono::duration<double, std::milli>(t_hash_done - t_hash0).count(), host_threads(nb),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_hash_done).count());
        }
        g0 = g1;
    }
    for (size_t i = 0; i < n; ++i) {                          // z on the domain, or more than 4096 elements: one polynomial at a time
        if (!fallback[i]) continue;
        const size_t np = blob_padded_len(lens[i]);
        PolySet& set = ctx->poly[0];
        void* d_evals = nullptr;
        int32_t rc = blob_to_fr_run(ctx, blobs[i], lens[i], np, &d_evals, ctx->stream, &set.c, &set.a);
        if (rc == KZG_OK) rc = proof_run(ctx, nullptr, nullptr, np, zs + 4 * i, nullptr, nullptr, ys + 4 * i, false, 0, nullptr);
        if (rc != KZG_OK) return rc;
    }
    return KZG_OK;
}

// batch.rs:76-168 on the host: 40 + 8 n + 128 n transcript bytes, one SHA-256, n - 1 field multiplications
void r_powers_host(const uint64_t* commitments, const uint64_t* zs, const uint64_t* ys, const uint64_t* proofs, const uint64_t* lens_elems,
                   size_t n, uint64_t* out) {
    using namespace kzg_host;
    std::vector<uint8_t> data(40 + n * 8 + n * 128, 0);
    memcpy(data.data(), RC_BATCH_DOMAIN, 24);                                          // bytes 24..31 stay zero (batch.rs:107-112)
    for (int b = 0; b < 8; ++b) data[32 + b] = (uint8_t)((uint64_t)n >> (8 * (7 - b)));
    for (size_t i = 0; i < n; ++i) for (int b = 0; b < 8; ++b) data[40 + 8 * i + b] = (uint8_t)(lens_elems[i] >> (8 * (7 - b)));
    uint8_t* rows = data.data() + 40 + 8 * n;
    const size_t per = 64, jobs = (n + per - 1) / per;
    parallel_for(jobs, [&](size_t j) {
        for (size_t i = j * per; i < std::min(n, (j + 1) * per); ++i) {
            uint8_t* p = rows + 128 * i;
            g1_serialize_compressed_ark(g1_from_wire(commitments + 8 * i), p);
            fr_wire_to_be_bytes(zs + 4 * i, p + 32);
            fr_wire_to_be_bytes(ys + 4 * i, p + 64);
            g1_serialize_compressed_ark(g1_from_wire(proofs + 8 * i), p + 96);
        }
    });
    Sha256 sh;
    sha256_init(sh);
    sha256_update(sh, data.data(), data.size());
    uint8_t dig[32];
    sha256_final(sh, dig);
    uint64_t r[4], cur[4];
    digest_to_fr_wire(dig, r);
    const uint64_t one_int[4] = {1, 0, 0, 0};
    fr_mul(FR_R2_WORDS, one_int, cur);                                                 // 1 in wire form
    for (size_t i = 0; i < n; ++i) { memcpy(out + 4 * i, cur, 32); fr_mul(cur, r, cur); }   // compute_powers (helpers.rs:298-313)
}

}  // namespace

int32_t kzg_compute_challenges_and_evaluate_polynomial(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens,
                                                       const uint64_t* commitments_xy_mont, size_t n, uint64_t* out_zs_mont, uint64_t* out_ys_mont) {
    if (!ctx) return KZG_ERR_INVALID_ARG;
    if (n == 0) return KZG_OK;
    if (!blobs || !blob_lens || !commitments_xy_mont || !out_zs_mont || !out_ys_mont) return KZG_ERR_INVALID_ARG;
    return challenges_and_evaluations(ctx, blobs, blob_lens, commitments_xy_mont, n, false, out_zs_mont, out_ys_mont);
}

int32_t kzg_evaluate_blobs_in_evaluation_form_batch(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens, const uint64_t* zs_mont,
                                                    size_t n, uint64_t* out_ys_mont) {
    if (!ctx) return KZG_ERR_INVALID_ARG;
    if (n == 0) return KZG_OK;
    if (!blobs || !blob_lens || !zs_mont || !out_ys_mont) return KZG_ERR_INVALID_ARG;
    return challenges_and_evaluations(ctx, blobs, blob_lens, nullptr, n, true, const_cast<uint64_t*>(zs_mont), out_ys_mont);
}

int32_t kzg_compute_r_powers(const uint64_t* commitments_xy_mont, const uint64_t* zs_mont, const uint64_t* ys_mont, const uint64_t* proofs_xy_mont,
                             const uint64_t* blobs_as_field_elements_length, size_t n, uint64_t* out_r_powers_mont) {
    if (n == 0) return KZG_OK;
    if (!commitments_xy_mont || !zs_mont || !ys_mont || !proofs_xy_mont || !blobs_as_field_elements_length || !out_r_powers_mont) return KZG_ERR_INVALID_ARG;
    r_powers_host(commitments_xy_mont, zs_mont, ys_mont, proofs_xy_mont, blobs_as_field_elements_length, n, out_r_powers_mont);
    return KZG_OK;
}

// verify::verify_blob_kzg_proof (verifier/src/verify.rs:76-98) in one call: validate both points, z = compute_challenge(blob, commitment),
// y = p(z) (one batched-evaluation launch of one blob), then verify::verify_proof (verify.rs:10-72) on the host.
int32_t kzg_verify_blob_kzg_proof(kzg_ctx* ctx, const uint8_t* blob_bytes, size_t len, const uint64_t commitment_xy_mont[8],
                                  const uint64_t proof_xy_mont[8], const uint64_t* g2_tau_mont, int32_t* out_ok) {
    if (!ctx || !out_ok || !commitment_xy_mont || !proof_xy_mont || (len && !blob_bytes)) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    if (!g1_on_curve(g1_from_wire(commitment_xy_mont)) || !g1_on_curve(g1_from_wire(proof_xy_mont))) return KZG_ERR_G1_NOT_ON_CURVE;   // verify.rs:82,85
    uint64_t z[4], y[4];
    const uint8_t* blobs[1] = {blob_bytes};
    const size_t lens[1] = {len};
    int32_t rc = challenges_and_evaluations(ctx, blobs, lens, commitment_xy_mont, 1, true, z, y);     // verify.rs:88-94
    if (rc != KZG_OK) return rc;
    return kzg_verify_proof(commitment_xy_mont, proof_xy_mont, y, z, g2_tau_mont, out_ok);            // verify.rs:97
}

int32_t kzg_verify_blob_kzg_proof_batch(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens, const uint64_t* commitments_xy_mont,
                                        const uint64_t* proofs_xy_mont, size_t n, const uint64_t* g2_tau_mont, int32_t* out_ok) {
    if (!ctx || !out_ok) return KZG_ERR_INVALID_ARG;
    if (n && (!blobs || !blob_lens || !commitments_xy_mont || !proofs_xy_mont)) return KZG_ERR_INVALID_ARG;
    using namespace kzg_host;
    const bool trace = opts().vb_trace != 0;                                                                   // phase times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_start = now();
    // batch.rs:29-37: every commitment, then every proof, on the curve (cofactor 1: no subgroup check to make) -- before anything else
    {
        std::atomic<int> bad{0};
        const size_t per = 128, jobs = (2 * n + per - 1) / per;           // (one pool job per point was 8 192 contended counter increments: 0.3 ms for 0.05 ms of work)
        parallel_for(jobs, [&](size_t j) {
            for (size_t k = j * per; k < std::min(2 * n, (j + 1) * per); ++k) {
                const uint64_t* p = k < n ? commitments_xy_mont + 8 * k : proofs_xy_mont + 8 * (k - n);
                if (!g1_on_curve(g1_from_wire(p))) bad.store(1, std::memory_order_relaxed);
            }
        });
        if (bad.load()) return KZG_ERR_G1_NOT_ON_CURVE;
    }
    const auto t_valid = now();
    std::vector<uint64_t> zs(4 * n), ys(4 * n), rp(4 * n), lens_elems(n);
    auto t_eval = t_valid, t_rp = t_valid;
    if (n) {
        int32_t rc = challenges_and_evaluations(ctx, blobs, blob_lens, commitments_xy_mont, n, true, zs.data(), ys.data());   // batch.rs:43-44
        if (rc != KZG_OK) return rc;
        t_eval = now();
        for (size_t i = 0; i < n; ++i) lens_elems[i] = (uint64_t)blob_padded_len(blob_lens[i]);                              // batch.rs:48-54
        RoctxRange range_rp("kzg:batch_verify:r_powers (host)");
        r_powers_host(commitments_xy_mont, zs.data(), ys.data(), proofs_xy_mont, lens_elems.data(), n, rp.data());           // batch.rs:222
        t_rp = now();
    }
    RoctxRange range_core("kzg:batch_verify:lincombs (GPU) + pairing (host)");
    const int32_t rc = verify_batch_core(ctx, commitments_xy_mont, zs.data(), ys.data(), proofs_xy_mont, rp.data(), n, g2_tau_mont, out_ok);   // batch.rs:62-68
    if (trace)
        fprintf(stderr, "kzg_verify_blob_kzg_proof_batch n=%zu: point validation %.3f ms, challenges + evaluations %.3f ms, r_powers %.3f ms, lincombs + pairing %.3f ms, call %.3f ms\n",
                n, ms(t_start, t_valid), ms(t_valid, t_eval), ms(t_eval, t_rp), ms(t_rp, now()), ms(t_start, now()));
    return rc;
}

}  // extern "C"

namespace kzg {
int32_t srs_upload_plain(kzg_ctx* ctx, const uint64_t* g1_xy_mont, size_t n_points, kzg_srs** out) { return srs_upload_impl(ctx, g1_xy_mont, n_points, out, false); }
}  // namespace kzg
