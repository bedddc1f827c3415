/*
 * kzg_bn254_mi355x.h — C-ABI of the MI355X-native KZG-BN254 prover hot path.
 *
 * Drop-in boundary for Layr-Labs/rust-kzg-bn254: the reference has no FFI seam of its own; its hot
 * path is five calls into arkworks 0.5.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference).  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add.
 *
 * Wire format (zero-copy from Rust):
 *   Fr / Fq      4 x u64 little-endian limbs, Montgomery form a * 2^256 mod m, canonical (< m) —
 *                arkworks' in-memory `Fp256<MontBackend<_, 4>>` (`fr.0.0`, cf. primitives/src/helpers.rs:158)
 *   G1 affine    8 x u64 = x[4] || y[4]; the identity is all-zero (the coordinates arkworks'
 *                `G1Affine::identity()` carries) and is also reported through `out_is_infinity`
 *   G1 partial   16 x u64 = X || Y || ZZ || ZZZ (extended Jacobian, x = X/ZZ, y = Y/ZZZ; identity: ZZ = 0)
 *
 * All pointers are caller-owned; the library never frees caller memory.  Functions return a
 * kzg_status (0 = OK, < 0 = error mirroring a `KzgError` variant, primitives/src/errors.rs:32-86).
 * A context is bound to one GPU; calls on one context are serialised internally, different
 * contexts may be used concurrently from different threads.  There is NO CPU fallback: without a
 * usable HIP device kzg_ctx_create fails with KZG_ERR_NO_DEVICE.
 *
 * ENVIRONMENT.  The library reads eleven variables and no others (csrc/engine.h `kzg::Opts`):
 *   KZG_NO_PRECOMPUTE=1, KZG_NO_NAF=1     SRS uploads without any tables / without the per-bit tables (read at every upload)
 *   KZG_HOST_THREADS_MAX, KZG_HOST_THREADS cap of (default min(48, hardware threads, the cgroup's CPU quota)) / exact width of the host pool
 *   KZG_VB_GROUP_BYTES, KZG_VB_CHUNK_BYTES, KZG_VB_TRACE   batch verification: bytes per GPU round / per upload chunk; phase times on stderr
 *   KZG_ROCTX=1                           roctx ranges around the host phases (rocprofv3 --marker-trace)
 *   KZG_NTT_TILE_LOG=10 / 11              one tile size of the Fr NTT at every transform size
 *   KZG_EXCHANGE_TIMEOUT_S, KZG_RCCL_LIB  the _rccl entries: seconds to wait for a collective (60); the RCCL to dlopen
 * Per-context settings are calls (kzg_ctx_set_msm_window, kzg_ctx_set_reduction_lanes).  Entry points marked MEASUREMENT ONLY below
 * (kzg_ctx_set_profiling, kzg_ctx_get_msm_profile*, kzg_ctx_measure_valu_rates) exist for bench.py's roofline figures; a host binding can leave them out.
 */
#ifndef KZG_BN254_MI355X_H
#define KZG_BN254_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kzg_ctx kzg_ctx;
typedef struct kzg_srs kzg_srs;

typedef enum {
    KZG_OK = 0,
    KZG_ERR_INVALID_ARG = -1,            /* null pointer / bad size */
    KZG_ERR_NO_DEVICE = -2,              /* no HIP device: the product never falls back to the CPU */
    KZG_ERR_DEVICE = -3,                 /* HIP runtime error, text in kzg_ctx_last_error */
    KZG_ERR_MSM_LENGTH_MISMATCH = -4,    /* arkworks msm Err(min_len) -> KzgError::MsmError / CommitError (helpers.rs:332, kzg.rs:102,123) */
    KZG_ERR_SRS_CAPACITY_EXCEEDED = -5,  /* KzgError::SrsCapacityExceeded (kzg.rs:89-94) */
    KZG_ERR_POLY_LENGTH = -6,            /* SerializationError("polynomial length is not correct") (kzg.rs:112-116) */
    KZG_ERR_NOT_POWER_OF_TWO = -7,       /* FFTError("length provided is not a power of 2") (kzg.rs:265-269) */
    KZG_ERR_DOMAIN = -8,                 /* domain construction failed: n > 2^28 (polynomial.rs:132-134, kzg.rs:276-278) */
    KZG_ERR_ROOTS_LENGTH = -9,           /* GenericError("inconsistent length between blob and root of unities") (kzg.rs:135-139,222-226) */
    KZG_ERR_INVALID_INPUT_LENGTH = -10,  /* KzgError::InvalidInputLength (helpers.rs:485-487) */
    KZG_ERR_TOO_LARGE = -11,             /* GenericError("Input size exceeds maximum polynomial size") (polynomial.rs:42-46) */
    KZG_ERR_ROOT_NOT_FOUND = -12,        /* GenericError("Root of unity not found") (kzg.rs:199-201) */
    KZG_ERR_ZERO_LENGTH = -13,           /* GenericError("Length of data after padding is 0") (helpers.rs:554-558) */
    KZG_ERR_SRS_LENGTH = -14,            /* GenericError("the length of data after padding is not valid with respect to the SRS") (helpers.rs:560-566) */
    KZG_ERR_DESERIALIZE = -15,           /* DeserializationError("point at infinity not coded properly for g1") (helpers.rs:191-195) */
    KZG_ERR_NOT_ON_CURVE = -16,          /* NotOnCurveError("compressed g1 point not on curve: ..") (helpers.rs:203-208) */
    KZG_ERR_G1_NOT_ON_CURVE = -17,       /* NotOnCurveError("G1 point not on curve") (helpers.rs:694-699, validate_g1_point) */
    KZG_ERR_G2_TAU_NOT_ON_CURVE = -18,   /* NotOnCurveError("Invalid trusted setup: G2_TAU not on curve") (verify.rs:29-33, batch.rs:214-216) */
    KZG_ERR_TAU_EQUALS_Z = -19,          /* GenericError("Evaluation point equals trusted setup secret") (verify.rs:56-60) */
    /* multi-rank calls only (no counterpart in the single-process reference): */
    KZG_ERR_PEER = -20,                  /* another rank of the communicator failed in this collective call (it returns its own error) */
    KZG_ERR_EXCHANGE_TIMEOUT = -21,      /* the all-gather did not complete within KZG_EXCHANGE_TIMEOUT_S (default 60): a peer is gone */
    KZG_ERR_IO = -22                     /* kzg_srs_save_packed / kzg_srs_load_packed: the file could not be opened, read or written */
} kzg_status;

/* The reference's error string for a status (Appendix B of SURVEY.md). */
const char* kzg_status_message(int32_t status);

/* ---- context -------------------------------------------------------------------------------- */
int32_t kzg_device_count(void);
int32_t kzg_ctx_create(int32_t device_id, kzg_ctx** out);
void    kzg_ctx_destroy(kzg_ctx* ctx);
const char* kzg_ctx_last_error(const kzg_ctx* ctx);
/* Tunables (0 = automatic): MSM window bits c in [2,16]; accumulate segment length.  A non-zero c also forces the
 * generic (no precomputed tables) mode for SRS-based calls. */
int32_t kzg_ctx_set_msm_window(kzg_ctx* ctx, int32_t c_bits, int32_t segment_len);
/* Lanes per point of the table-mode bucket-reduction kernels: 0 = automatic (lane quads for an MSM that runs alone, lane pairs
 * beside another MSM in flight), 2 = always pairs, 4 = always quads.  Results are identical; a test / measurement hook. */
int32_t kzg_ctx_set_reduction_lanes(kzg_ctx* ctx, int32_t lanes);
/* MEASUREMENT ONLY (bench.py's roofline): when enabled, every MSM launch is bracketed phase by phase with HIP events on the
 * context's launch stream.  phase_ms_out[0..7] = accumulated milliseconds of: digits, bucket scan, scatter,
 * segment map, bucket ACCUMULATE (the dominant kernel), bucket finalise, window reduction, whole device span;
 * *launches / *pairs = launches and (scalar, point) pairs covered.  Enabling resets the counters. */
int32_t kzg_ctx_set_profiling(kzg_ctx* ctx, int32_t enable);
int32_t kzg_ctx_get_msm_profile(kzg_ctx* ctx, double phase_ms_out[8], uint64_t* launches, uint64_t* pairs);
/* *entries = sorted entries (= mixed additions of the accumulate kernel) of the launches profiled since kzg_ctx_set_profiling(ctx, 1):
 * windows x pairs with the fixed-window tables, data dependent (~254 / (w + 1) + 1/2 per scalar) in the NAF mode of the per-bit tables. */
int32_t kzg_ctx_get_msm_profile_entries(kzg_ctx* ctx, uint64_t* entries);
/* MEASUREMENT ONLY: issue rate of the instruction classes the MSM accumulate kernel consists of, on this device, with
 * `waves_per_simd` waves per SIMD (the kernel runs 3): out_ns[0..5] = nanoseconds per wave-instruction per SIMD of
 * v_mad_i64_i32, v_mul_lo_u32, v_ashrrev_i64, v_and_b32, v_sub_u32 (the plain 32-bit class), s_nop.  ~60 ms of GPU time. */
int32_t kzg_ctx_measure_valu_rates(kzg_ctx* ctx, int32_t waves_per_simd, double out_ns[6]);

/* ---- SRS: device-resident monomial G1 powers -------------------------------------------------- */
/* Replaces holding `SRS.g1: Cow<[G1Affine]>` (prover/src/srs.rs:11-21) on the host and copying
 * `srs.g1[..n].to_vec()` on every commit (kzg.rs:119).  n points, 8 u64 each.  Uploaded once; for
 * 128 <= n (and table size <= 48 GiB) the upload also precomputes the window tables 2^(c w) * P_i
 * (ceil(255 / c) x 64 B per point, c = clamp(floor(log2 n) - 4, 7, 16); KZG_NO_PRECOMPUTE=1 disables). */
int32_t kzg_srs_upload(kzg_ctx* ctx, const uint64_t* g1_xy_mont, size_t n_points, kzg_srs** out);
/* SRS::new's point decoding on the GPU (prover/src/srs.rs:35-49, :51-68 -> helpers.rs:175-226 read_g1_point_from_bytes_be):
 * `bytes` = the first n_points x 32 bytes of a gnark-format compressed G1 file (big-endian x, flags in the top two bits of
 * byte 0).  All points are decompressed in one kernel (one 254-bit exponentiation per point) and the SRS is made resident
 * like kzg_srs_upload.  A malformed point -> KZG_ERR_DESERIALIZE / KZG_ERR_NOT_ON_CURVE with *bad_index = its position
 * (the reference's loader panics there, srs.rs:60-63). */
int32_t kzg_srs_load_compressed_be(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index);
/* Test / bench utility (no counterpart in the reference, which loads ceremony files): synthetic SRS with a
 * KNOWN tau, P_i = tau^(first_power + i) * G1, generated on the device (a shard of the powers when
 * first_power > 0); and read-back of a resident SRS in wire format. */
int32_t kzg_srs_generate(kzg_ctx* ctx, const uint64_t tau_mont[4], uint64_t first_power, size_t n_points, kzg_srs** out);
int32_t kzg_srs_download(kzg_ctx* ctx, const kzg_srs* srs, size_t offset, size_t n, uint64_t* out_xy_mont);
/* A loaded SRS written once in the library's own packed form and read back WITHOUT decoding (the reference decodes the ceremony
 * file at every SRS::new, prover/src/srs.rs:35-188 -- one field exponentiation per point; its README quotes "a few minutes" for the mainnet file):
 * "KZGSRS1\0" | u64 n | u64 0 | SHA-256 of the payload | n x 64 B points exactly as kzg_srs_upload takes them.  Loading verifies the digest
 * (KZG_ERR_DESERIALIZE: wrong magic, truncated, trailing bytes, digest mismatch) and, on the device, that every point is on the curve or the
 * identity (KZG_ERR_NOT_ON_CURVE), then builds the tables like kzg_srs_upload.  points_to_load = 0: all of them; more than the file holds ->
 * KZG_ERR_SRS_LENGTH; a file that cannot be opened / written -> KZG_ERR_IO. */
int32_t kzg_srs_save_packed(kzg_ctx* ctx, const kzg_srs* srs, const char* path);
int32_t kzg_srs_load_packed(kzg_ctx* ctx, const char* path, size_t points_to_load, kzg_srs** out);
void    kzg_srs_free(kzg_srs* srs);
size_t  kzg_srs_len(const kzg_srs* srs);
/* The same loader for the "native" format of SRS::parallel_read_g1_points_native(.., is_native = true) (prover/src/srs.rs:205-251 ->
 * primitives/src/traits.rs:34-36: G1Affine::deserialize_compressed of ark-serialize 0.5): x as 32 little-endian bytes, flags in the top
 * two bits of the LAST byte (0x80 = the larger y, 0x40 = infinity; both set, or x >= p -> KZG_ERR_DESERIALIZE).  RESTATED from
 * ark-serialize, not pinned by a reference vector: the reference tree holds no file in this format and calls the function nowhere. */
int32_t kzg_srs_load_compressed_ark_le(kzg_ctx* ctx, const uint8_t* bytes, size_t n_points, kzg_srs** out, uint64_t* bad_index);
/* 1 when the SRS carries its per-bit tables Bit_p[i] = 2^p P_i (the NAF mode of MSMs of >= 2^14 pairs and the batched launches need
 * them; absent with KZG_NO_NAF=1, above 2^22 points, or when HBM is short).  build != 0: try to build them first (once, on the SRS's
 * own context).
 *
 * THREADS.  An SRS handle may be used from any number of host threads and by every context of its GPU at once, as the reference shares
 * one `SRS` across threads (prover/tests/kzg_test.rs:9-17, primitives/tests/blob_test.rs:83-94).  Tables that are built after upload
 * -- these per-bit tables on the first batched call of a small SRS, the x3 / x5 / x7 tables of kzg_g1_ifft(64..2048) (<= 100 MB, built on the first such call), the bases attached by
 * kzg_srs_cache_lagrange -- are built under a lock of the SRS and published only when complete; calls on ONE context are serialised by
 * the context's own lock, so N threads sharing a context see N serial calls; threads that need to overlap use one context each
 * (contexts of one GPU share the SRS).  kzg_srs_free / kzg_srs_drop_lagrange must not race with calls that use the handle.
 * tests/test_gpu_concurrency.py exercises exactly this contract. */
int32_t kzg_srs_has_bit_tables(kzg_srs* srs, int32_t build);

/* ---- arkworks boundary 1: <G1Projective as VariableBaseMSM>::msm(bases, scalars).into_affine() -- */
/* Call sites: prover/src/kzg.rs:100, :121; primitives/src/helpers.rs:332 (g1_lincomb).
 * n_bases != n_scalars -> KZG_ERR_MSM_LENGTH_MISMATCH (arkworks returns Err(min_len)). */
int32_t kzg_msm_g1(kzg_ctx* ctx, const uint64_t* bases_xy_mont, size_t n_bases,
                   const uint64_t* scalars_mont, size_t n_scalars,
                   uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* `batch` independent MSMs of n pairs each in one kernel sequence: bases = batch x n points, scalars = batch x n
 * elements (MSM b uses the b-th block of each), out = batch x 8 u64 (+ batch infinity flags).  This is the shape of
 * verifier/src/batch.rs:228,245,246 (three g1_lincomb calls of equal length); results equal `batch` kzg_msm_g1 calls. */
int32_t kzg_msm_g1_batch(kzg_ctx* ctx, const uint64_t* bases_xy_mont, const uint64_t* scalars_mont, size_t n, size_t batch,
                         uint64_t* out_xy_mont, uint8_t* out_is_infinity);
/* Same with bases = srs[offset .. offset + n) already resident (commit path). */
int32_t kzg_msm_g1_srs(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                       const uint64_t* scalars_mont, size_t n,
                       uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* Scalars already in device memory (d_scalars_mont: device pointer to n x 4 u64). */
int32_t kzg_msm_g1_srs_device(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                              const void* d_scalars_mont, size_t n,
                              uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* Multi-GPU sharding: partial sum of one shard, not converted to affine (16 u64 XYZZ). */
int32_t kzg_msm_g1_srs_partial_device(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                                      const void* d_scalars_mont, size_t n, uint64_t out_xyzz_mont[16]);
int32_t kzg_msm_g1_srs_partial(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                               const uint64_t* scalars_mont, size_t n, uint64_t out_xyzz_mont[16]);
#ifndef KZG_NUM_SLOTS
#define KZG_NUM_SLOTS 4
#endif
/* Asynchronous form for streams of commitments (software pipelining): `begin` enqueues the whole kernel sequence of one
 * MSM over srs[offset .. offset + n) on the stream of `slot` (0 .. KZG_NUM_SLOTS-1; each slot has its own stream and workspace) and
 * returns without waiting; `end` waits for that slot, runs the O(1) host epilogue and writes the affine point
 * (out_xy_mont / out_is_infinity) and/or the unconverted partial sum (out_xyzz_mont, 16 u64); either may be NULL.
 * With begin(k+1) issued before end(k), the sort / bucket-reduction phases of one MSM run beside the accumulation of the
 * other and the host epilogue leaves the critical path.  d_scalars_mont must be complete before `begin` and stay
 * untouched until `end`.  1 <= n <= 2^26 (at most 64 launches: an SRS of more than 2^20 points is served in launches of 2^20 pairs;
 * the synchronous calls have no cap; the asynchronous commit_eval_form / commit_blob / compute_proof forms below stop at 2^24);
 * a slot that is still in flight (or, for `end`, idle) -> KZG_ERR_INVALID_ARG.
 * The synchronous MSM / commit / proof calls use slot 0's workspace: while slot 0 is in flight they return
 * KZG_ERR_INVALID_ARG (text in kzg_ctx_last_error); the other slots may be in flight beside them.  Two slots hide the latency-bound
 * phases of a 2^20-pair MSM; shard-sized MSMs (2^17 .. 2^18 pairs per GPU) keep gaining up to four. */
int32_t kzg_msm_g1_srs_device_begin(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                                    const void* d_scalars_mont, size_t n, int32_t slot);
/* Same with the scalars in host memory (n x 4 u64): the H2D copy is issued on the slot's stream into the slot's staging
 * buffer, so with two slots alternating it runs beside the other slot's kernels.  The host buffer may be reused as soon as
 * `begin` returns only if it is pageable memory (the runtime stages it); a pinned buffer must stay valid until `end`. */
int32_t kzg_msm_g1_srs_begin(kzg_ctx* ctx, const kzg_srs* srs, size_t offset,
                             const uint64_t* scalars_mont, size_t n, int32_t slot);
int32_t kzg_msm_g1_srs_end(kzg_ctx* ctx, int32_t slot, uint64_t* out_xy_mont, uint8_t* out_is_infinity,
                           uint64_t* out_xyzz_mont);
/* Fold `count` gathered partial sums (count x 16 u64) and convert to affine.  Host-only, O(count). */
int32_t kzg_g1_fold_partials(const uint64_t* partials_xyzz_mont, size_t count,
                             uint64_t out_xy_mont[8], uint8_t* out_is_infinity);

/* ---- arkworks boundary 2: GeneralEvaluationDomain::<Fr>::new(n).{fft,ifft}(&[Fr]) --------------- */
/* Call sites: primitives/src/polynomial.rs:131-135 (ifft), :242-246 (fft).  Natural order in and out,
 * omega = 5^((r-1)/n); the inverse transform includes the n^-1 scaling.  In place.
 * n not a power of two -> KZG_ERR_NOT_POWER_OF_TWO; n > 2^28 -> KZG_ERR_DOMAIN. */
int32_t kzg_fr_ntt(kzg_ctx* ctx, uint64_t* data_mont, size_t n, int32_t inverse);
int32_t kzg_fr_ntt_device(kzg_ctx* ctx, void* d_data_mont, size_t n, int32_t inverse);

/* ---- KZG surface (prover/src/kzg.rs) ----------------------------------------------------------- */
/* KZG::commit_coeff_form (kzg.rs:107-125): n > srs len -> KZG_ERR_POLY_LENGTH. */
int32_t kzg_commit_coeff_form(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* coeffs_mont, size_t n,
                              uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* Batched forms (no counterpart in the reference, which commits one polynomial per call): `count` polynomials of n coefficients
 * (evaluations) each, one after the other in memory, against the first n points of ONE SRS -> count commitments, out_xy_mont = count x 8
 * words, out_is_infinity = count bytes (may be NULL).  Same values and errors as count calls of kzg_commit_coeff_form /
 * kzg_commit_eval_form; ONE kernel sequence per 1 024 polynomials (width-8 NAF digits over the SRS's per-bit tables, 64 buckets per
 * polynomial): a few microseconds per commitment of 512..2048 coefficients instead of 0.14-0.2 ms.  The SRS handle is not const:
 * an SRS of fewer than 2^15 points gets its per-bit tables (16 KiB per point) on the first batched call, and the eval form keeps
 * the Lagrange basis of n points with the SRS (kzg_srs_cache_lagrange).  _device: the scalars are already in device memory. */
int32_t kzg_commit_coeff_form_batch(kzg_ctx* ctx, kzg_srs* srs, const uint64_t* coeffs_mont, size_t n, size_t count,
                                    uint64_t* out_xy_mont, uint8_t* out_is_infinity);
int32_t kzg_commit_coeff_form_batch_device(kzg_ctx* ctx, kzg_srs* srs, const void* d_coeffs_mont, size_t n, size_t count,
                                           uint64_t* out_xy_mont, uint8_t* out_is_infinity);
int32_t kzg_commit_eval_form_batch(kzg_ctx* ctx, kzg_srs* srs, const uint64_t* evals_mont, size_t n, size_t count,
                                   uint64_t* out_xy_mont, uint8_t* out_is_infinity);
int32_t kzg_commit_eval_form_batch_device(kzg_ctx* ctx, kzg_srs* srs, const void* d_evals_mont, size_t n, size_t count,
                                          uint64_t* out_xy_mont, uint8_t* out_is_infinity);
/* ONE batched launch, asynchronous (the building block of the calls above, for streams of resident scalar sets -- e.g. the ranks of a
 * sharded commitment, which group several steps into one launch): `count` scalar sets of n scalars each, in SEPARATE device buffers,
 * against srs[offset .. offset + n) on `slot`; count <= min(16, kzg_msm_batch_capacity(n)).  _end_batch waits and returns count affine
 * points (count x 8 words) and / or count XYZZ partials (count x 16 words), in order. */
size_t kzg_msm_batch_capacity(size_t n);
int32_t kzg_msm_g1_srs_device_begin_batch(kzg_ctx* ctx, kzg_srs* srs, size_t offset, const void* const* d_scalars_mont, size_t n, size_t count, int32_t slot);
int32_t kzg_msm_g1_srs_end_batch(kzg_ctx* ctx, int32_t slot, size_t count, uint64_t* out_xy_mont, uint8_t* out_is_infinity, uint64_t* out_xyzz_mont);
/* KZG::commit_eval_form (kzg.rs:84-104): n > srs len -> KZG_ERR_SRS_CAPACITY_EXCEEDED; n not a power of
 * two -> KZG_ERR_NOT_POWER_OF_TWO.  Computed as MSM(srs, IFFT(evals)), identical to the reference's
 * MSM(g1_ifft(srs), evals) (prover/src/lib.rs:43-47; prover/tests/kzg_test.rs:57-89). */
int32_t kzg_commit_eval_form(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                             uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* helpers::to_fr_array + PolynomialEvalForm::new padding (helpers.rs:40-57, polynomial.rs:41-57): each 32-byte big-endian
 * chunk of the (already padded) blob mod r, last chunk right-padded with zeros, zero-extended to the next power of two;
 * *n_out = that length; out needs cap >= *n_out elements (call with out = NULL to query). */
int32_t kzg_blob_to_fr(kzg_ctx* ctx, const uint8_t* blob_bytes, size_t len, uint64_t* out_mont, size_t cap, size_t* n_out);
/* KZG::commit_blob (kzg.rs:182-185) = Blob::to_polynomial_eval_form + commit_eval_form, bytes in, point out. */
int32_t kzg_commit_blob(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len,
                        uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
/* Asynchronous forms of the two calls above for streams of blobs: the whole chain (H2D copy, bytes -> Fr, IFFT, MSM) is
 * enqueued on the stream of `slot`; collect the commitment with kzg_msm_g1_srs_end(ctx, slot, ...).  Host input
 * buffers follow the rule of kzg_msm_g1_srs_begin.  Same error codes as the synchronous calls; n (resp. the padded blob
 * length) <= 2^24. */
int32_t kzg_commit_eval_form_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n, int32_t slot);
int32_t kzg_commit_blob_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, int32_t slot);
/* KZG::g1_ifft (kzg.rs:263-285): Lagrange-basis SRS L_i = n^-1 sum_j w^(-ij) P_j of the first n SRS points, natural
 * order, n x 8 u64 written to out.  n not a power of two -> KZG_ERR_NOT_POWER_OF_TWO ("length provided is not a
 * power of 2"); n > 2^28 -> KZG_ERR_DOMAIN.  Not used by the commit / proof path of this library. */
int32_t kzg_g1_ifft(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint64_t* out_xy_mont);
/* The same Lagrange basis kept ON THE DEVICE.  The reference recomputes g1_ifft inside every commit_eval_form
 * (prover/src/kzg.rs:96-98); here it is computed once per (SRS, n):
 *   kzg_srs_cache_lagrange  builds it (with its MSM window tables) and attaches it to `srs`; from then on kzg_commit_eval_form of
 *                           exactly n evaluations is ONE MSM over it (the reference's literal form, kzg.rs:98-100) instead of
 *                           IFFT + MSM over the monomial basis -- the same point either way;
 *   kzg_srs_lagrange        returns it as an SRS handle of its own (free with kzg_srs_free): kzg_msm_g1_srs over it with the
 *                           evaluations as scalars is commit_eval_form;
 *   kzg_srs_drop_lagrange   releases what kzg_srs_cache_lagrange attached.
 * Errors as kzg_g1_ifft. */
int32_t kzg_srs_cache_lagrange(kzg_ctx* ctx, kzg_srs* srs, size_t n);
int32_t kzg_srs_lagrange(kzg_ctx* ctx, const kzg_srs* srs, size_t n, kzg_srs** out);
int32_t kzg_srs_drop_lagrange(kzg_ctx* ctx, kzg_srs* srs);
/* KZG::compute_proof / compute_proof_impl (kzg.rs:128-178, :215-234, on-domain branch :237-260).
 * roots = KZG::expanded_roots_of_unity (n_roots entries); n != n_roots -> KZG_ERR_ROOTS_LENGTH.
 * out_y (optional, 4 u64) receives y = p(z) (helpers.rs:475-535). */
int32_t kzg_compute_proof(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                          const uint64_t* roots_mont, size_t n_roots, const uint64_t z_mont[4],
                          uint64_t out_xy_mont[8], uint8_t* out_is_infinity, uint64_t* out_y_mont);
/* Asynchronous form of kzg_compute_proof for streams of proofs (same slots as kzg_msm_g1_srs_begin): upload, batch inversion,
 * y, quotient, IFFT and MSM are all enqueued on the slot's stream; `end` waits and returns the proof point and y = p(z). */
int32_t kzg_compute_proof_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* evals_mont, size_t n,
                                const uint64_t* roots_mont, size_t n_roots, const uint64_t z_mont[4], int32_t slot);
int32_t kzg_compute_proof_end(kzg_ctx* ctx, int32_t slot, uint64_t out_xy_mont[8], uint8_t* out_is_infinity,
                              uint64_t* out_y_mont);
/* Multi-GPU forms of the two calls above (BASELINE config 4).  srs_shard holds the SRS powers
 * [shard_lo, shard_lo + kzg_srs_len(srs_shard)); every rank passes the whole polynomial, performs the O(n) field work
 * (IFFT, quotient) redundantly and commits only its slice of the coefficients; out = the unconverted partial sum (16 u64)
 * that the ranks all-gather and fold with kzg_g1_fold_partials.  The shards must jointly cover [0, n). */
int32_t kzg_commit_eval_form_partial(kzg_ctx* ctx, const kzg_srs* srs_shard, size_t shard_lo,
                                     const uint64_t* evals_mont, size_t n, uint64_t out_xyzz_mont[16]);
int32_t kzg_compute_proof_partial(kzg_ctx* ctx, const kzg_srs* srs_shard, size_t shard_lo,
                                  const uint64_t* evals_mont, size_t n, const uint64_t* roots_mont, size_t n_roots,
                                  const uint64_t z_mont[4], uint64_t out_xyzz_mont[16], uint64_t* out_y_mont);
/* ---- BASELINE config 4 sharded by EVALUATION index over the Lagrange basis (csrc/lagrange.hip) --------------------------
 * The reference commits an evaluation-form polynomial as MSM(g1_ifft(srs), evals) (prover/src/kzg.rs:96-100) and a proof as the same MSM
 * over the evaluations of the quotient q_i = (f_i - y) / (w^i - z) (kzg.rs:151-177; the entry of a domain point z = w^m: kzg.rs:237-260).
 * Both are sums over the evaluation index, so a rank that holds the Lagrange points L_i and the evaluations f_i of a contiguous index
 * range [lo, lo + len) needs nothing else: it uploads, inverts and divides its OWN slice (4 MiB at n = 2^20 over 8 ranks) -- no replicated
 * upload, no IFFT, no whole-polynomial quotient (the kzg_*_partial forms above do all three on every rank).  Two small exchanges per proof:
 *     phase 1   S_g = sum_{i in slice} f_i w^i / (z - w^i)   ->  all-gather of G x 64 B  ->  y = (z^n - 1) / n sum_g S_g   (helpers.rs:507-532;
 *               z = w^m: y = f_m, sent by the slice that owns m, helpers.rs:497-504)
 *     phase 2   q_i on the slice, MSM over the slice          ->  all-gather of G x 256 B ->  proof = fold of the partial points (+ q_m L_m)
 * Results are the SAME field elements and the same affine point as kzg_compute_proof / kzg_commit_eval_form on one GPU.
 *
 * kzg_srs_slice            a copy of srs[lo, lo + len) as an SRS handle of its own (its own window / per-bit tables)
 * kzg_srs_lagrange_shard   KZG::g1_ifft(n) of the first n points of `srs` (kzg.rs:263-285), of which the points [lo, lo + len) are kept as a
 *                          handle of their own: this rank's shard of the Lagrange basis (one-time set-up; errors as kzg_g1_ifft).  A host that
 *                          already holds the Lagrange points uploads its slice with kzg_srs_upload instead. */
int32_t kzg_srs_slice(kzg_ctx* ctx, const kzg_srs* srs, size_t lo, size_t len, kzg_srs** out);
int32_t kzg_srs_lagrange_shard(kzg_ctx* ctx, const kzg_srs* srs, size_t n, size_t lo, size_t len, kzg_srs** out);
/* KZG::commit_eval_form (kzg.rs:84-104) of this rank's slice: sum_{i in slice} f_i L_i, unconverted (16 words, folded with
 * kzg_g1_fold_partials).  len > kzg_srs_len(lagrange_shard) -> KZG_ERR_SRS_CAPACITY_EXCEEDED. */
int32_t kzg_commit_eval_form_lagrange_partial(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const uint64_t* evals_slice_mont, size_t len,
                                              uint64_t out_xyzz_mont[16]);
int32_t kzg_commit_eval_form_lagrange_partial_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const void* d_evals_slice_mont, size_t len,
                                                     uint64_t out_xyzz_mont[16]);
/* KZG::compute_proof_impl (kzg.rs:128-178) on a slice, in four steps on `slot` (the asynchronous slots of kzg_msm_g1_srs_begin; several
 * proofs may be in flight on different slots):
 *   _begin       enqueue: upload of the slice (or, _device, the caller's resident buffer read in place: keep it untouched until _end),
 *                the inverses 1 / (w^i - z) of the slice (one inversion per 1 024 elements, side by side; they need z only and run BESIDE the upload)
 *                and S_g.  A pageable host buffer may be reused when _begin returns (the runtime has staged it); a PINNED one must stay valid and
 *                unchanged until _partial_y has returned, as for kzg_msm_g1_srs_begin.  shard_lo = index of the
 *                slice's first evaluation; shard_lo + len <= n, len <= kzg_srs_len(lagrange_shard), n a power of two <= 2^28.
 *   _partial_y   waits for phase 1: out_ypart = 8 words, S_g | f_m (f_m only from the slice that owns m when z = w^m, else zero)
 *   kzg_lagrange_fold_y        (host only) y from the G gathered 8-word parts
 *   _continue    enqueue: y goes up, q_i on the slice, the slice's MSM over the Lagrange shard
 *   _end         waits: out_part = 32 words: [0,16) XYZZ partial | [16,20) T_g = sum_{i != m} q_i w^i (z = w^m only) | [20,28) L_m, [28] = 1
 *                (the owner of m only) | zeros
 *   kzg_lagrange_fold_proof    (host only) the proof point from the G gathered 32-word parts (z = w^m: + q_m L_m, q_m = -(1/z) sum_g T_g)
 *   _abort       gives up whatever the slot has in flight (a peer failed between two phases)
 * An empty slice (len = 0) is valid in every step and contributes zeros / the identity.  A step called out of order, or on a slot that is
 * busy otherwise -> KZG_ERR_INVALID_ARG. */
#define KZG_LAGRANGE_YPART_WORDS 8
#define KZG_LAGRANGE_PART_WORDS 32
int32_t kzg_compute_proof_lagrange_begin(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont,
                                         size_t len, size_t n, const uint64_t z_mont[4], int32_t slot);
int32_t kzg_compute_proof_lagrange_begin_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont,
                                                size_t len, size_t n, const uint64_t z_mont[4], int32_t slot);
/* Commitment AND proof of one slice, for streams of blobs (BASELINE config 4 per rank, several blobs in flight): as _begin on `proof_slot`, and the
 * slice's commitment sum_i f_i L_i is enqueued on `commit_slot` from the SAME uploaded copy (one H2D for both; _device: both read the caller's buffer).
 * Collect the commitment's partial with kzg_msm_g1_srs_end(ctx, commit_slot, NULL, NULL, out_xyzz) (len = 0: nothing was enqueued there, the partial
 * is the identity); proof_slot must not be begun again before that.  The lagrange_shard handle and (_device) the evaluations must stay valid and
 * untouched until _end / _abort of the proof and the end of the commitment. */
/* commit_slot == proof_slot: GROUPED -- the two MSMs of the blob (the evaluations and the quotient over the same shard) leave as ONE batched launch
 * of the MSM engine from _continue (shard-sized MSMs are bound by dependent latency: two per launch cost 0.33 ms where two launches cost 0.46 at 2^17 pairs);
 * one slot per blob, so four blobs may be in flight.  Needs the shard's per-bit tables (kzg_srs_has_bit_tables(shard, 1) builds them for a shard of
 * fewer than 2^11 points) and kzg_msm_batch_capacity(len) >= 2 (slices of up to 2^18 elements), else KZG_ERR_INVALID_ARG.  Collect BOTH partial sums with kzg_commit_and_prove_lagrange_end (kzg_compute_proof_lagrange_end refuses a
 * grouped slot). */
int32_t kzg_commit_and_prove_lagrange_end(kzg_ctx* ctx, int32_t slot, uint64_t out_commit_xyzz_mont[16], uint64_t out_part[32]);
int32_t kzg_commit_and_prove_lagrange_begin(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont, size_t len,
                                            size_t n, const uint64_t z_mont[4], int32_t commit_slot, int32_t proof_slot);
int32_t kzg_commit_and_prove_lagrange_begin_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont, size_t len,
                                                   size_t n, const uint64_t z_mont[4], int32_t commit_slot, int32_t proof_slot);
int32_t kzg_compute_proof_lagrange_partial_y(kzg_ctx* ctx, int32_t slot, uint64_t out_ypart_mont[8]);
int32_t kzg_compute_proof_lagrange_continue(kzg_ctx* ctx, int32_t slot, const uint64_t y_mont[4]);
int32_t kzg_compute_proof_lagrange_end(kzg_ctx* ctx, int32_t slot, uint64_t out_part[32]);
int32_t kzg_compute_proof_lagrange_abort(kzg_ctx* ctx, int32_t slot);
int32_t kzg_lagrange_fold_y(const uint64_t* yparts_mont, size_t count, size_t n, const uint64_t z_mont[4], uint64_t out_y_mont[4]);
/* KZG::compute_quotient_eval_on_domain (prover/src/kzg.rs:237-260): sum over the n roots w^i != z of (f_i - value) w^i / ((z - w^i) z), i.e. the
 * quotient's evaluation AT the domain point z = w^m when value = f_m; like the reference, z need not be a domain point (then no term is skipped).
 * n evaluations in host memory, n a power of two (else KZG_ERR_NOT_POWER_OF_TWO; n > 2^28 -> KZG_ERR_DOMAIN); z = 0 -> KZG_ERR_INVALID_ARG (the
 * reference divides by z).  Computed on the GPU by the kernels of the Lagrange-sharded proof (csrc/lagrange.hip) on slot 0, which must be idle. */
int32_t kzg_compute_quotient_eval_on_domain(kzg_ctx* ctx, const uint64_t z_mont[4], const uint64_t* evals_mont, size_t n, const uint64_t value_mont[4],
                                            uint64_t out_quotient_mont[4]);
int32_t kzg_lagrange_fold_proof(const uint64_t* parts, size_t count, size_t n, const uint64_t z_mont[4], uint64_t out_xy_mont[8],
                                uint8_t* out_is_infinity);
/* ---- several GPUs behind one handle (no torch, no RCCL) ------------------------------------------------------
 * One context, one resident SRS shard and one host thread per entry of device_ids (an id may appear more than once: several
 * contexts on one GPU).  Device g holds the SRS powers [g N / G, (g+1) N / G) and commits that slice of every polynomial; the G
 * partial sums are folded on the host.  Same results, same status codes as the single-GPU calls they mirror:
 *   kzg_multi_commit_coeff_form  KZG::commit_coeff_form  (prover/src/kzg.rs:107-125)
 *   kzg_multi_commit_eval_form   KZG::commit_eval_form   (kzg.rs:84-104; every device transforms the whole polynomial)
 *   kzg_multi_compute_proof      KZG::compute_proof      (kzg.rs:215-234; the O(n) field work is done redundantly per device) */
typedef struct kzg_multi kzg_multi;
int32_t kzg_multi_create(const int32_t* device_ids, int32_t n_devices, kzg_multi** out);
void kzg_multi_destroy(kzg_multi* m);
int32_t kzg_multi_device_count(const kzg_multi* m);
size_t kzg_multi_srs_len(const kzg_multi* m);
int32_t kzg_multi_srs_upload(kzg_multi* m, const uint64_t* g1_xy_mont, size_t n_points);
int32_t kzg_multi_srs_generate(kzg_multi* m, const uint64_t tau_mont[4], size_t n_points);        /* known-tau SRS (tests, bench) */
/* One-time set-up for BASELINE config 4 WITHOUT replicated work: the Lagrange basis of the first n powers (KZG::g1_ifft, kzg.rs:263-285),
 * sharded by evaluation index (device g keeps L_i, i in [g n / G, (g+1) n / G), with its own tables).  From then on
 * kzg_multi_commit_eval_form / kzg_multi_compute_proof of exactly n evaluations make every device read, invert and divide ITS slice of the
 * caller's buffer only (csrc/lagrange.hip): no whole-polynomial upload, IFFT or quotient per device.  Same results.  Replaced by the next
 * kzg_multi_srs_upload / _generate. */
int32_t kzg_multi_cache_lagrange(kzg_multi* m, size_t n);
int32_t kzg_multi_commit_coeff_form(kzg_multi* m, const uint64_t* coeffs_mont, size_t n, uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_multi_commit_eval_form(kzg_multi* m, const uint64_t* evals_mont, size_t n, uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_multi_compute_proof(kzg_multi* m, const uint64_t* evals_mont, size_t n, size_t n_roots, const uint64_t z_mont[4],
                                uint64_t out_xy_mont[8], uint8_t* out_is_infinity, uint64_t* out_y_mont);
/* Resident inputs and streams (a commitment service; what `bench.py --multi` times): kzg_multi_scalars_upload keeps the slice
 * [g N / G, (g+1) N / G) of polynomial `buffer_id` (0 .. 15, n coefficients) in the HBM of device g; kzg_multi_commit_resident_stream
 * commits `count` of the uploaded polynomials (buffer_ids[k]) -- every device runs its own software pipeline over the stream (two
 * or three MSMs in flight, the asynchronous slots of kzg_msm_g1_srs_device_begin), the count x G partial sums are folded on the
 * host -- and writes count affine points. */
int32_t kzg_multi_scalars_upload(kzg_multi* m, int32_t buffer_id, const uint64_t* coeffs_mont, size_t n);
int32_t kzg_multi_commit_resident_stream(kzg_multi* m, const int32_t* buffer_ids, size_t count, uint64_t* out_xy_mont,
                                         uint8_t* out_is_infinity);

/* ---- one process per GPU: the exchanges over RCCL, behind the C-ABI (BASELINE north_star) -----------------------------------
 * RCCL has no reduction operator for elliptic-curve addition, so the "all-reduce" of the G partial sums is ONE all-gather of G small rows
 * over xGMI and a fold of G points on every rank.  Every rank of the communicator must make the same call (they are collectives).
 *   kzg_rccl_allgather_fold        partial in -> the same folded affine point on every rank
 *   kzg_commit_coeff_form_rccl     KZG::commit_coeff_form (prover/src/kzg.rs:107-125): this rank's resident coefficient slice over its shard of the
 *                                  monomial SRS + the exchange + the fold
 *   kzg_commit_eval_form_rccl      KZG::commit_eval_form (kzg.rs:84-104), BASELINE config 4's commitment: this rank's slice of the EVALUATIONS
 *                                  over its shard of the Lagrange basis (kzg_srs_lagrange_shard) + the exchange + the fold
 *   kzg_compute_proof_rccl         KZG::compute_proof_impl (kzg.rs:128-178, :237-260), config 4's proof, from the same slice: two exchanges
 *                                  (G x 72 B: partial barycentric sums -> y;  G x 264 B: partial points) -- the four steps of
 *                                  kzg_compute_proof_lagrange_* with the collectives in between; out_y (optional) = p(z)
 *   (_device: the slice is already in device memory and is read in place)
 * nccl_comm = the caller's ncclComm_t for this context's device.  The library uses the RCCL that is already in the process (the one that
 * made the communicator), else loads /opt/rocm/lib/librccl.so (KZG_RCCL_LIB overrides); no link-time dependency.  `world` must equal
 * ncclCommCount(comm) (checked: KZG_ERR_INVALID_ARG).
 * FAILURES.  A rank whose local work fails still issues every collective of the call with a poisoned row and then returns its own
 * status; every other rank returns KZG_ERR_PEER (kzg_ctx_last_error names the failed ranks): no rank is left blocked.  A peer that never
 * arrives: the wait is bounded by KZG_EXCHANGE_TIMEOUT_S seconds (default 60) -> KZG_ERR_EXCHANGE_TIMEOUT; exit, the communicator is
 * unusable.  Python hosts use torch.distributed instead (sharding.py: the same rows through PyTorch's communicator); one process driving
 * all GPUs uses kzg_multi_* (no collective). */
int32_t kzg_rccl_allgather_fold(kzg_ctx* ctx, void* nccl_comm, int32_t world, const uint64_t partial_xyzz_mont[16],
                                uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_commit_coeff_form_rccl(kzg_ctx* ctx, const kzg_srs* srs_shard, const void* d_coeffs_shard_mont, size_t n_shard, void* nccl_comm,
                                   int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_commit_eval_form_rccl(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const uint64_t* evals_slice_mont, size_t len, void* nccl_comm,
                                  int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_commit_eval_form_rccl_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, const void* d_evals_slice_mont, size_t len, void* nccl_comm,
                                         int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity);
int32_t kzg_compute_proof_rccl(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const uint64_t* evals_slice_mont, size_t len, size_t n,
                               const uint64_t z_mont[4], void* nccl_comm, int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity,
                               uint64_t* out_y_mont);
int32_t kzg_compute_proof_rccl_device(kzg_ctx* ctx, const kzg_srs* lagrange_shard, size_t shard_lo, const void* d_evals_slice_mont, size_t len, size_t n,
                                      const uint64_t z_mont[4], void* nccl_comm, int32_t world, uint64_t out_xy_mont[8], uint8_t* out_is_infinity,
                                      uint64_t* out_y_mont);

/* helpers::validate_g1_point (helpers.rs:694-708): KZG_OK, or KZG_ERR_G1_NOT_ON_CURVE (y^2 != x^3 + 3; the identity = all zeros is accepted, the
 * cofactor of G1 is 1 so there is no subgroup check to make).  Host-only. */
int32_t kzg_validate_g1_point(const uint64_t xy_mont[8]);
/* helpers::hash_to_field_element (helpers.rs:382-390): SHA-256 of msg, read big-endian, mod r.  Host-only. */
int32_t kzg_hash_to_field_element(const uint8_t* msg, size_t len, uint64_t out_mont[4]);
/* NOTE on the two Fiat-Shamir transcripts below (kzg_compute_challenge, kzg_compute_r_powers): their byte layout follows the reference
 * line by line, but the 32-byte compressed G1 encoding inside them (x little-endian, 0x80 = larger y, 0x40 = infinity) is ark-serialize's
 * `serialize_compressed`, RESTATED here and in oracle/ alike and pinned by NO vector the reference holds -- its own tests only check that the
 * functions are deterministic (primitives/tests/helpers_test.rs:846-949).  Everything else in this header is pinned by reference fixtures.
 *
 * helpers::compute_challenge (primitives/src/helpers.rs:411-472): the Fiat-Shamir evaluation point of a blob,
 *   z = SHA-256( "EIGENDA_FSBLOBVERIFY_V1_" || u64be(n) || n x 32 B evaluations (big-endian, canonical) || commitment ) mod r,
 * n = next_pow2(ceil(len / 32)); evaluations = the blob's 32-byte big-endian chunks mod r (Blob::to_polynomial_eval_form, last
 * chunk right-padded with zeros, zero elements up to n); commitment = ark-serialize compressed G1Affine (32 B, x little-endian,
 * flags in the top bits of the last byte).  Host only (x86 SHA extensions when present); the commitment must be on the curve
 * (helpers::validate_g1_point, helpers.rs:694-708): else KZG_ERR_G1_NOT_ON_CURVE. */
int32_t kzg_compute_challenge(const uint8_t* blob_bytes, size_t len, const uint64_t commitment_xy_mont[8], uint64_t out_z_mont[4]);
/* KZG::compute_blob_proof (prover/src/kzg.rs:288-309): validate the commitment, z = compute_challenge(blob, commitment),
 * proof = compute_proof_impl(blob.to_polynomial_eval_form(), z, srs).  n_roots = KZG::expanded_roots_of_unity.len()
 * (KZG_ERR_ROOTS_LENGTH when it differs from the padded blob length, kzg.rs:135-139).  Bytes in, point out: the blob crosses
 * PCIe once; its transcript hash runs on a host thread beside the upload / bytes->Fr kernels.  out_z_mont / out_y_mont
 * (optional) receive the challenge and p(z). */
int32_t kzg_compute_blob_proof(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                               const uint64_t commitment_xy_mont[8], uint64_t out_proof_xy_mont[8], uint8_t* out_is_infinity,
                               uint64_t* out_z_mont, uint64_t* out_y_mont);
/* KZG::commit_blob (kzg.rs:182-185) followed by KZG::compute_blob_proof (kzg.rs:288-309) on the same blob in one call: the
 * transcript prefix (tag, length, evaluations: everything but the commitment) is hashed on a host thread WHILE the GPU computes
 * the commitment, and finalised with the 32 commitment bytes; then the proof.  Same results as the two separate calls. */
int32_t kzg_commit_and_prove_blob(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                                  uint64_t out_commitment_xy_mont[8], uint8_t* out_commitment_is_infinity,
                                  uint64_t out_proof_xy_mont[8], uint8_t* out_proof_is_infinity,
                                  uint64_t* out_z_mont, uint64_t* out_y_mont);
/* The same as a STREAM (round 6): KZG::commit_blob + KZG::compute_blob_proof (kzg.rs:182-185, :288-309) of many blobs with up to
 * KZG_BLOB_JOBS of them in flight per context.  One call above is bound by ITS transcript -- SHA-256 over the whole blob is one
 * sequential stream, 14-15 ms per 32 MiB on a core with SHA extensions, against ~2.9 ms of GPU work -- but the transcripts of different
 * blobs are independent:
 *   _begin(job)  starts the job's transcript prefix (helpers.rs:411-455: everything but the commitment) on a host thread of its own and
 *                enqueues upload, bytes -> Fr and the commitment on one of the context's slots; returns without waiting for either.
 *                commitment_xy_mont != NULL: KZG::compute_blob_proof as it stands -- the caller's commitment is validated (kzg.rs:295,
 *                KZG_ERR_G1_NOT_ON_CURVE) and absorbed, none is computed.
 *   _end(job)    returns the job's commitment, proof, z and y (any out pointer may be NULL); waits for whatever is still missing.
 * EVERY begin / end call also moves every other job on as far as it can go without waiting (commitment collected -> 32 bytes appended
 * to its finished prefix -> z -> proof enqueued), so with `begin(k + d); end(k)` in a loop the hashes of d blobs run side by
 * side and the GPU works through commitments and proofs back to back: 2.9-3.1 ms per 32 MiB blob at d = 12 (3.1 at d = 8) instead of 16-17.  Results are bit-identical to
 * kzg_commit_and_prove_blob / kzg_compute_blob_proof.  Same argument checks and statuses as those; blobs of up to 2^24 elements.
 * LIFETIME: blob_bytes is read by the transcript thread until the job's _end call returns.  SLOTS: the jobs take the context's
 * KZG_NUM_SLOTS slots phase by phase (a slot still held by another kzg_*_begin call of the caller is left alone; the synchronous calls,
 * which run on slot 0, return KZG_ERR_INVALID_ARG while a job holds it -- collect the jobs first).  A failed job reports its status from
 * _end and is idle afterwards.  kzg_ctx_destroy joins whatever is still in flight. */
#ifndef KZG_BLOB_JOBS
#define KZG_BLOB_JOBS 16
#endif
int32_t kzg_commit_and_prove_blob_begin(kzg_ctx* ctx, const kzg_srs* srs, const uint8_t* blob_bytes, size_t len, size_t n_roots,
                                        const uint64_t* commitment_xy_mont, int32_t job);
int32_t kzg_commit_and_prove_blob_end(kzg_ctx* ctx, int32_t job, uint64_t* out_commitment_xy_mont, uint8_t* out_commitment_is_infinity,
                                      uint64_t* out_proof_xy_mont, uint8_t* out_proof_is_infinity, uint64_t* out_z_mont, uint64_t* out_y_mont);
/* helpers::evaluate_polynomial_in_evaluation_form (helpers.rs:475-535) on the domain of size n. */
int32_t kzg_evaluate_polynomial_in_evaluation_form(kzg_ctx* ctx, const uint64_t* evals_mont, size_t n,
                                                   const uint64_t z_mont[4], uint64_t out_y_mont[4]);
/* helpers::calculate_roots_of_unity (helpers.rs:553-589): out receives next_pow2(ceil(len/32)) elements
 * (capacity `cap` elements); *n_out = count. */
int32_t kzg_calculate_roots_of_unity(kzg_ctx* ctx, uint64_t length_of_data_after_padding,
                                     uint64_t* out_mont, size_t cap, size_t* n_out);

/* ---- verifier surface ------------------------------------------------------------
 * G2 wire format: 16 u64 = x.c0 | x.c1 | y.c0 | y.c1, each a 4 x u64 Montgomery Fq (arkworks' Fq2{c0,c1} in memory);
 * identity = all zeros.  The pairing itself is O(1) per verification and runs on the host (optimal ate Miller loop + final
 * exponentiation, the construction arkworks computes; csrc/host_pairing.h); the n-point linear combinations and the n barycentric
 * evaluations of batch verification run on the GPU.  g2_tau_mont = NULL selects consts::G2_TAU (primitives/src/consts.rs:55-64). */
int32_t kzg_g2_generator(uint64_t out_g2_mont[16]);                  /* G2Affine::generator() */
int32_t kzg_g2_tau_mainnet(uint64_t out_g2_mont[16]);                /* consts::G2_TAU */
int32_t kzg_g2_mul_generator(const uint64_t scalar_mont[4], uint64_t out_g2_mont[16]);   /* [s]G2 (tests / custom setups) */
/* helpers::is_on_curve_g2 (helpers.rs:263-285): y^2 = x^3 + 3 / (9 + u) on the twist; the identity counts as on the curve.  Host-only. */
int32_t kzg_g2_is_on_curve(const uint64_t g2_mont[16], int32_t* out_on_curve);
/* helpers::example_validate_g2_point (helpers.rs:740-766), checks in the reference's order: *out_reason = 0 valid | 1 NotOnCurveError("G2 point not on
 * curve") | 2 NotOnCurveError("G2 point is point at infinity") | 3 NotOnCurveError("G2 point not in correct subgroup") ([r]P != O) |
 * 4 G2GeneratorNotAcceptedError("G2 point cannot be the generator point").  Host-only. */
int32_t kzg_validate_g2_point(const uint64_t g2_mont[16], int32_t* out_reason);
/* helpers::pairings_verify(a1, a2, b1, b2) (helpers.rs:392-398): *out_ok = (e(a1,a2) == e(b1,b2)).  Host-only. */
int32_t kzg_pairings_verify(const uint64_t a1_xy_mont[8], const uint64_t a2_g2_mont[16],
                            const uint64_t b1_xy_mont[8], const uint64_t b2_g2_mont[16], int32_t* out_ok);
/* verify::verify_proof (verifier/src/verify.rs:10-72).  Off-curve commitment/proof -> KZG_ERR_G1_NOT_ON_CURVE;
 * [tau - z]G2 == identity -> KZG_ERR_TAU_EQUALS_Z.  Host-only.
 * g2_tau (here and in kzg_verify_kzg_proof_batch): NULL = consts::G2_TAU; a caller-supplied point is only checked to be ON THE
 * CURVE, exactly as the reference does (verify.rs:29-33, batch.rs:212-214) -- its membership in the order-r subgroup of the twist
 * is NOT checked; outside that subgroup a pairing value is not defined by the reference either (a degenerate Miller-loop step,
 * T = +-Q, makes this library answer "not equal"): pass subgroup points (any [s]G2). */
int32_t kzg_verify_proof(const uint64_t commitment_xy_mont[8], const uint64_t proof_xy_mont[8],
                         const uint64_t value_mont[4], const uint64_t z_mont[4],
                         const uint64_t* g2_tau_mont, int32_t* out_ok);
/* verify::verify_blob_kzg_proof (verifier/src/verify.rs:76-98) in one call: both points validated (KZG_ERR_G1_NOT_ON_CURVE), the blob's
 * challenge z = compute_challenge(blob, commitment) (helpers.rs:411-472, host SHA-256), y = p(z) (helpers.rs:475-535, GPU), then
 * verify_proof above.  blob_bytes = the padded bytes of the blob (Blob::data()); an empty blob -> KZG_ERR_ZERO_LENGTH. */
int32_t kzg_verify_blob_kzg_proof(kzg_ctx* ctx, const uint8_t* blob_bytes, size_t len, const uint64_t commitment_xy_mont[8],
                                  const uint64_t proof_xy_mont[8], const uint64_t* g2_tau_mont, int32_t* out_ok);
/* batch::verify_kzg_proof_batch (verifier/src/batch.rs:185-256) after the caller has derived r_powers from its
 * transcript (compute_r_powers, batch.rs:76-168): point validation, the three g1_lincomb calls as one batched GPU MSM,
 * the final 2-pairing check. */
int32_t kzg_verify_kzg_proof_batch(kzg_ctx* ctx, const uint64_t* commitments_xy_mont, const uint64_t* zs_mont,
                                   const uint64_t* ys_mont, const uint64_t* proofs_xy_mont,
                                   const uint64_t* r_powers_mont, size_t n,
                                   const uint64_t* g2_tau_mont, int32_t* out_ok);
/* batch::compute_r_powers (verifier/src/batch.rs:76-168): r = SHA-256("EIGENDA_RCKZGBATCH___V1_" || 8 zero bytes || u64be(n) ||
 * n x u64be(blobs_as_field_elements_length[i]) || n x (C_i || z_i || y_i || proof_i)) mod r with the points ark-compressed and z, y
 * canonical big-endian; out = [r^0 .. r^(n-1)] (helpers::compute_powers, helpers.rs:298-313).  Host only (rows serialised on a
 * thread pool). */
int32_t kzg_compute_r_powers(const uint64_t* commitments_xy_mont, const uint64_t* zs_mont, const uint64_t* ys_mont,
                             const uint64_t* proofs_xy_mont, const uint64_t* blobs_as_field_elements_length, size_t n,
                             uint64_t* out_r_powers_mont);
/* helpers::compute_challenges_and_evaluate_polynomial (primitives/src/helpers.rs:613-662) for n blobs in ONE call:
 * out_zs[i] = compute_challenge(blob_i, commitment_i) (helpers.rs:411-472), out_ys[i] = p_i(z_i) (helpers.rs:475-535).
 * blobs[i] / blob_lens[i] = the padded bytes of blob i (Blob::data()).  The n transcripts are hashed on a pool of host threads
 * (KZG_HOST_THREADS_MAX, default min(48, hardware threads, the cgroup's CPU quota)); the n barycentric evaluations run as one batched GPU launch for blobs of up to
 * 4096 field elements (one workgroup per blob, one inversion per blob) and through the single-polynomial path beyond.  Errors, for
 * the first failing blob in order: KZG_ERR_TOO_LARGE (polynomial.rs:42-46), KZG_ERR_G1_NOT_ON_CURVE (helpers.rs:413),
 * KZG_ERR_ZERO_LENGTH (empty blob: helpers.rs:554-558). */
int32_t kzg_compute_challenges_and_evaluate_polynomial(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens,
                                                       const uint64_t* commitments_xy_mont, size_t n,
                                                       uint64_t* out_zs_mont, uint64_t* out_ys_mont);
/* The evaluation half of the call above with the points given: out_ys[i] = p_i(zs[i]) for n blobs (helpers.rs:475-535 each, incl.
 * the early return for z on the domain, helpers.rs:497-504), one batched GPU launch.  KZG_ERR_TOO_LARGE / KZG_ERR_ZERO_LENGTH as above. */
int32_t kzg_evaluate_blobs_in_evaluation_form_batch(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens,
                                                    const uint64_t* zs_mont, size_t n, uint64_t* out_ys_mont);
/* batch::verify_blob_kzg_proof_batch (verifier/src/batch.rs:16-69) end to end -- BASELINE config 5 in one call: validation of the
 * 2n points (KZG_ERR_G1_NOT_ON_CURVE), the n challenges and evaluations (above), compute_r_powers, then verify_kzg_proof_batch
 * (three batched GPU MSMs + the host 2-pairing check).  *out_ok = 1 iff every proof verifies.  n = 0 -> *out_ok = 1. */
int32_t kzg_verify_blob_kzg_proof_batch(kzg_ctx* ctx, const uint8_t* const* blobs, const size_t* blob_lens,
                                        const uint64_t* commitments_xy_mont, const uint64_t* proofs_xy_mont, size_t n,
                                        const uint64_t* g2_tau_mont, int32_t* out_ok);

#ifdef __cplusplus
}
#endif
#endif
