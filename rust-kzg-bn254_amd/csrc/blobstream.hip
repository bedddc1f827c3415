// blobstream.hip — blob -> commitment + proof as a STREAM of jobs (prover/src/kzg.rs:182-185 + :288-309 on many blobs).
//
// One call of KZG::compute_blob_proof is bound by its Fiat-Shamir transcript: SHA-256 over tag || n || 32 n bytes || commitment
// (primitives/src/helpers.rs:411-472) is ONE sequential hash stream, 14-15 ms for a 32 MiB blob on a core with SHA extensions, against
// ~2.9 ms of GPU work for commitment + proof.  The hashes of DIFFERENT blobs are independent, so a job here is
//     begin:  transcript prefix on a host thread of its own  |  upload, bytes -> Fr, commitment MSM on a slot of the context
//     ...     (the caller begins further jobs: their hashes run side by side, the GPU works through commitments and proofs)
//     pump:   commitment collected -> prefix joined, 32 commitment bytes appended -> z -> proof enqueued on a free slot
//     end:    proof collected
// with KZG_BLOB_JOBS jobs in flight per context.  The jobs own their device buffers (bytes, evaluations); the MSM / polynomial
// workspaces are those of the context's KZG_NUM_SLOTS slots, taken per phase.  Every call advances every job that can advance
// without waiting ("pump"), so proofs of earlier jobs are enqueued behind the commitment of the job just begun; the transcript thread
// pumps too when it finishes (if the context is not busy in another call), and an end call waits OUTSIDE the context's lock, so a
// proof is enqueued the moment its challenge exists and not at the caller's next call.
#include "engine.h"
#include "host_curve.h"
#include "host_pairing.h"
#include "host_sha256.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <pthread.h>
#include <thread>

namespace kzg {

// capi.hip
void challenge_absorb_prefix(kzg_host::Sha256& sh, const uint8_t* blob, size_t len, size_t n_padded);
void challenge_finish(kzg_host::Sha256& sh, const kzg_host::G1& commitment, uint64_t out_z_mont[4]);
size_t blob_padded_len(size_t len);
// poly.hip
int32_t blob_to_fr_run(kzg_ctx* ctx, const uint8_t* bytes, size_t len, size_t n_padded, void** d_out, hipStream_t st, DeviceBuffer* d_bytes,
                       DeviceBuffer* d_elems);
int32_t proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals, size_t n, const uint64_t z[4], int slot, const void* d_resident);
int32_t proof_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_y);

enum BlobJobState { JOB_IDLE = 0, JOB_COMMIT, JOB_WAIT_HASH, JOB_PROOF, JOB_DONE, JOB_FAILED };

struct BlobJob {
    int state = JOB_IDLE;
    uint64_t seq = 0;                  // order of the begin calls: the oldest job in flight is the one collected when a slot is needed
    const kzg_srs* srs = nullptr;
    size_t n = 0;
    DeviceBuffer d_bytes, d_evals;
    hipEvent_t ev_evals = nullptr;     // behind bytes -> Fr on the stream of the begin call: the proof (maybe another stream) starts after it
    std::thread hasher;
    std::atomic<int> hash_done{0};
    kzg_host::Sha256 sh;
    int slot = -1;                     // slot of the phase in flight (JOB_COMMIT, JOB_PROOF)
    kzg_host::G1 commitment;
    uint64_t cxy[8] = {};
    uint8_t cinf = 0;
    uint64_t z[4] = {}, y[4] = {}, pxy[8] = {};
    uint8_t pinf = 0;
    int32_t rc = KZG_OK;
};

struct BlobStream {
    BlobJob job[KZG_BLOB_JOBS];
    uint64_t next_seq = 1;
    int owner[KZG_NUM_SLOTS];          // job holding the slot, or -1
    bool closing = false;              // kzg_ctx_destroy is under way: transcript threads no longer pump (nothing new may reach the streams)
    BlobStream() { for (int& o : owner) o = -1; }
};

static bool slot_is_free(const kzg_ctx* ctx, const BlobStream& bs, int s) {
    return bs.owner[s] < 0 && !ctx->slot_pending[s] && ctx->lag[s].phase == 0;
}

static void job_fail(BlobStream& bs, BlobJob& j, int32_t rc) {
    if (j.slot >= 0) { bs.owner[j.slot] = -1; j.slot = -1; }
    j.rc = rc;
    j.state = JOB_FAILED;
}

// the phase in flight on the job's slot has finished on the device?
static bool job_phase_finished(kzg_ctx* ctx, const BlobJob& j) {
    hipEvent_t ev = ctx->slot_msm(j.slot).ev_done;
    if (!ev) return true;
    const hipError_t q = hipEventQuery(ev);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return false; }
    return true;                       // finished, or an error that the collecting call will report
}

// waits for the phase in flight and moves the job on: commitment -> JOB_WAIT_HASH, proof -> JOB_DONE
static void job_collect(kzg_ctx* ctx, BlobStream& bs, BlobJob& j) {
    const int s = j.slot;
    if (j.state == JOB_COMMIT) {
        const int32_t rc = msm_end(ctx, s, j.cxy, &j.cinf, nullptr);
        bs.owner[s] = -1; j.slot = -1;
        if (rc != KZG_OK) { job_fail(bs, j, rc); return; }
        j.commitment = kzg_host::g1_from_wire(j.cxy);
        j.state = JOB_WAIT_HASH;
    } else if (j.state == JOB_PROOF) {
        const int32_t rc = proof_end(ctx, s, j.pxy, &j.pinf, j.y);
        bs.owner[s] = -1; j.slot = -1;
        if (rc != KZG_OK) { job_fail(bs, j, rc); return; }
        j.state = JOB_DONE;
    }
}

// a free slot; when there is none, the oldest job holding one is collected.  -1: every slot is held by other asynchronous calls of the caller
static int acquire_slot(kzg_ctx* ctx, BlobStream& bs) {
    for (;;) {
        for (int s = 0; s < KZG_NUM_SLOTS; ++s) if (slot_is_free(ctx, bs, s)) return s;
        BlobJob* oldest = nullptr;
        for (BlobJob& j : bs.job) if (j.slot >= 0 && (!oldest || j.seq < oldest->seq)) oldest = &j;
        if (!oldest) return -1;
        job_collect(ctx, bs, *oldest);
    }
}

// JOB_WAIT_HASH with the prefix hashed: z, then the proof on `slot`
static void job_start_proof(kzg_ctx* ctx, BlobStream& bs, BlobJob& j, int slot) {
    if (j.hasher.joinable() && j.hasher.get_id() != std::this_thread::get_id()) j.hasher.join();   // (the transcript thread itself may be the one pumping)
    challenge_finish(j.sh, j.commitment, j.z);
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc == KZG_OK && hipStreamWaitEvent(st, j.ev_evals, 0) != hipSuccess) rc = set_error(ctx, hipGetLastError(), "hipStreamWaitEvent(blob job)");
    if (rc == KZG_OK) rc = proof_begin(ctx, j.srs, nullptr, j.n, j.z, slot, j.d_evals.p);
    if (rc != KZG_OK) { job_fail(bs, j, rc); return; }
    j.slot = slot;
    bs.owner[slot] = (int)(&j - bs.job);
    j.state = JOB_PROOF;
}

// everything that can move without waiting: finished phases are collected, proofs whose challenge is ready go onto free slots (oldest first)
static void pump(kzg_ctx* ctx, BlobStream& bs) {
    for (BlobJob& j : bs.job)
        if ((j.state == JOB_COMMIT || j.state == JOB_PROOF) && job_phase_finished(ctx, j)) job_collect(ctx, bs, j);
    for (;;) {
        BlobJob* next = nullptr;
        for (BlobJob& j : bs.job)
            if (j.state == JOB_WAIT_HASH && j.hash_done.load(std::memory_order_acquire) && (!next || j.seq < next->seq)) next = &j;
        if (!next) return;
        int slot = -1;
        for (int s = 0; s < KZG_NUM_SLOTS && slot < 0; ++s) if (slot_is_free(ctx, bs, s)) slot = s;
        if (slot < 0) return;
        job_start_proof(ctx, bs, *next, slot);
    }
}

static void job_reset(BlobJob& j) {
    if (j.hasher.joinable()) j.hasher.join();
    j.state = JOB_IDLE;
    j.srs = nullptr;
    j.slot = -1;
    j.rc = KZG_OK;
    j.hash_done.store(0, std::memory_order_relaxed);
}

// kzg_ctx_destroy, BEFORE it drains the streams: from here on no transcript thread enqueues anything; they are joined (they read the callers' blobs),
// then the context's streams are synchronised and the buffers go
void blob_stream_release(kzg_ctx* ctx) {
    BlobStream* bs = ctx->blob_stream;
    if (!bs) return;
    { std::lock_guard<std::mutex> lk(ctx->mu); bs->closing = true; }
    for (BlobJob& j : bs->job) if (j.hasher.joinable()) j.hasher.join();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto st : ctx->stream_x) if (st) (void)hipStreamSynchronize(st);
    for (BlobJob& j : bs->job) {
        if (j.hasher.joinable()) j.hasher.join();
        if (j.ev_evals) (void)hipEventDestroy(j.ev_evals);
        j.d_bytes.release();
        j.d_evals.release();
    }
    delete bs;
    ctx->blob_stream = nullptr;
}

}  // namespace kzg

using namespace kzg;

extern "C" {

int32_t kzg_commit_and_prove_blob_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                                        const uint64_t* commitment_xy_mont, int32_t job) {
    if (!ctx || !srs || srs->ctx != ctx || (len && !blob_bytes) || job < 0 || job >= KZG_BLOB_JOBS) return KZG_ERR_INVALID_ARG;
    const size_t n_elems = (len + 31) / 32;
    if (n_elems > ((size_t)1 << 24)) return KZG_ERR_TOO_LARGE;                       // (the asynchronous forms stop at 2^24 elements)
    const size_t n = blob_padded_len(len);
    kzg_host::G1 given;
    if (commitment_xy_mont) {
        given = kzg_host::g1_from_wire(commitment_xy_mont);
        if (!kzg_host::g1_on_curve(given)) return KZG_ERR_G1_NOT_ON_CURVE;           // kzg.rs:295
    }
    if (n != n_roots) return KZG_ERR_ROOTS_LENGTH;                                   // kzg.rs:135-139
    if (n > srs->n) return KZG_ERR_SRS_CAPACITY_EXCEEDED;                            // kzg.rs:89-94
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->blob_stream) {
        ctx->blob_stream = new (std::nothrow) BlobStream();
        if (!ctx->blob_stream) return KZG_ERR_DEVICE;
    }
    BlobStream& bs = *ctx->blob_stream;
    BlobJob& j = bs.job[job];
    if (j.state != JOB_IDLE) { ctx->last_error = "this blob job is in flight: call kzg_commit_and_prove_blob_end first"; return KZG_ERR_INVALID_ARG; }
    pump(ctx, bs);
    const int slot = acquire_slot(ctx, bs);
    if (slot < 0) { ctx->last_error = "no slot free: every slot is held by another kzg_*_begin call"; return KZG_ERR_INVALID_ARG; }
    hipStream_t st = nullptr;
    int32_t rc = msm_slot_stream(ctx, slot, &st);
    if (rc != KZG_OK) return rc;
    if (!j.ev_evals) KZG_HIP_TRY(ctx, hipEventCreateWithFlags(&j.ev_evals, hipEventDisableTiming));
    // the transcript prefix on a thread of its own (it reads the caller's buffer until the job's end call returns)
    kzg_host::sha256_init(j.sh);
    j.hash_done.store(0, std::memory_order_relaxed);
    BlobJob* jp = &j;
    // small blobs (the reference's bench sizes, 10-50 KB): the prefix costs less than starting a thread (~50 us) -- hashed here, before the enqueues
    constexpr size_t INLINE_HASH_MAX = (size_t)64 << 10;
    if (len <= INLINE_HASH_MAX) {
        challenge_absorb_prefix(j.sh, blob_bytes, len, n);
        j.hash_done.store(1, std::memory_order_release);
    } else
    try {
        j.hasher = std::thread([ctx, jp, blob_bytes, len, n] {
            pthread_setname_np(pthread_self(), "kzg-hash");
            challenge_absorb_prefix(jp->sh, blob_bytes, len, n);
            jp->hash_done.store(1, std::memory_order_release);
            // the proof goes out now if the context is free (an end call waits outside the lock); try_lock, never lock: a caller may be joining
            // this thread while it holds the lock, and whoever holds it pumps on its way out anyway
            if (ctx->mu.try_lock()) {
                try {
                    if (ctx->blob_stream && !ctx->blob_stream->closing && hipSetDevice(ctx->device) == hipSuccess) pump(ctx, *ctx->blob_stream);
                } catch (...) { }                                   // (an allocation failure here must not end the process: the next begin / end call pumps again and reports it)
                ctx->mu.unlock();
            }
        });
    } catch (...) {
        ctx->last_error = "could not start the transcript thread of a blob job";
        return KZG_ERR_DEVICE;
    }
    // from here on a failure has to wait for that thread (job_reset joins it): it reads the caller's buffer
    void* d_evals = nullptr;
    rc = blob_to_fr_run(ctx, blob_bytes, len, n, &d_evals, st, &j.d_bytes, &j.d_evals);              // Blob::to_polynomial_eval_form
    if (rc == KZG_OK && hipEventRecord(j.ev_evals, st) != hipSuccess) rc = set_error(ctx, hipGetLastError(), "hipEventRecord(blob job)");
    if (rc != KZG_OK) { job_reset(j); return rc; }
    j.srs = srs;
    j.n = n;
    j.seq = bs.next_seq++;
    if (commitment_xy_mont) {                                                        // compute_blob_proof: the commitment is the caller's
        j.commitment = given;
        memcpy(j.cxy, commitment_xy_mont, 64);
        j.cinf = given.inf ? 1 : 0;
        j.state = JOB_WAIT_HASH;
    } else {                                                                         // commit_blob (kzg.rs:182-185 -> :84-104)
        if (const kzg_srs* cached = srs_cached_lagrange(srs, n)) {                   // the evaluations are the scalars: no copy, no IFFT
            rc = msm_begin(ctx, slot, srs_bases(cached, 0, n, ctx->msm_c_override == 0), d_evals, n);
        } else {
            MsmWorkspace& ws = ctx->slot_msm(slot);
            hipError_t e = ws.scalars.reserve(n * 32 + 32);
            if (e == hipSuccess) e = hipMemcpyAsync(ws.scalars.p, d_evals, n * 32, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) rc = set_error(ctx, e, "blob job: copy of the evaluations");
            if (rc == KZG_OK && n > 1) {
                NttTables tb;
                int log_n = 0; while (((size_t)1 << log_n) < n) ++log_n;
                rc = ntt_get_tables(ctx, log_n, true, &tb);
                if (rc == KZG_OK) rc = ntt_run(ctx, ws.scalars.p, n, true, st, &ctx->slot_ntt(slot));
            }
            if (rc == KZG_OK) rc = msm_begin(ctx, slot, srs_bases(srs, 0, n, ctx->msm_c_override == 0), ws.scalars.p, n);
        }
        if (rc != KZG_OK) { (void)hipStreamSynchronize(st); job_reset(j); return rc; }
        j.slot = slot;
        bs.owner[slot] = job;
        j.state = JOB_COMMIT;
    }
    pump(ctx, bs);                                                                   // (a given commitment with its prefix already hashed: the proof goes out now)
    return KZG_OK;
}

int32_t kzg_commit_and_prove_blob_end(kzg_ctx* ctx, int32_t job, uint64_t* out_commitment_xy_mont, uint8_t* out_commitment_is_infinity,
                                      uint64_t* out_proof_xy_mont, uint8_t* out_proof_is_infinity, uint64_t* out_z_mont, uint64_t* out_y_mont) {
    if (!ctx || job < 0 || job >= KZG_BLOB_JOBS) return KZG_ERR_INVALID_ARG;
    for (;;) {
        hipEvent_t wait_event = nullptr;                 // what this call waits for next, outside the lock: a phase on the device ...
        BlobJob* wait_hash = nullptr;                    // ... or the job's transcript prefix
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
            if (!ctx->blob_stream || ctx->blob_stream->job[job].state == JOB_IDLE) {
                ctx->last_error = "no blob job in flight under this index";
                return KZG_ERR_INVALID_ARG;
            }
            BlobStream& bs = *ctx->blob_stream;
            BlobJob& j = bs.job[job];
            pump(ctx, bs);
            if (j.state == JOB_WAIT_HASH && j.hash_done.load(std::memory_order_acquire)) {
                // challenge ready but every slot taken: collect older jobs until one is free (acquire_slot only collects jobs that hold a slot: not this one)
                const int slot = acquire_slot(ctx, bs);
                if (slot < 0) { job_fail(bs, j, KZG_ERR_INVALID_ARG); ctx->last_error = "no slot free: every slot is held by another kzg_*_begin call"; }
                else if (j.state == JOB_WAIT_HASH) job_start_proof(ctx, bs, j, slot);
            }
            if (j.state == JOB_DONE) {
                if (out_commitment_xy_mont) memcpy(out_commitment_xy_mont, j.cxy, 64);
                if (out_commitment_is_infinity) *out_commitment_is_infinity = j.cinf;
                if (out_proof_xy_mont) memcpy(out_proof_xy_mont, j.pxy, 64);
                if (out_proof_is_infinity) *out_proof_is_infinity = j.pinf;
                if (out_z_mont) memcpy(out_z_mont, j.z, 32);
                if (out_y_mont) memcpy(out_y_mont, j.y, 32);
                job_reset(j);
                return KZG_OK;
            }
            if (j.state == JOB_FAILED) {
                const int32_t rc = j.rc;
                job_reset(j);
                return rc;
            }
            if (j.state == JOB_COMMIT || j.state == JOB_PROOF) wait_event = ctx->slot_msm(j.slot).ev_done;
            else wait_hash = &j;
            if ((j.state == JOB_COMMIT || j.state == JOB_PROOF) && !wait_event) { job_collect(ctx, bs, j); continue; }
        }
        // Outside the lock: the transcript threads of other jobs can pump meanwhile.  The event may be re-recorded for another phase if this job is
        // collected by such a pump -- then the wait is a little longer than needed and the state is looked at again.
        if (wait_event) {
            if (hipEventSynchronize(wait_event) != hipSuccess) (void)hipGetLastError();   // (an error surfaces in the collecting call under the lock)
        } else {
            while (!wait_hash->hash_done.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
}

}  // extern "C"
