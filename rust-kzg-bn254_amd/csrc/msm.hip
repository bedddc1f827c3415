// msm.hip — host orchestration of the G1 MSM kernels (msm_kernels.h).
// Replaces `G1Projective::msm(..)` + `.into_affine()` at prover/src/kzg.rs:100-101, :121-122 and
// primitives/src/helpers.rs:332-336.
//
// Two modes share the digit / sort / accumulate kernels:
//   table mode   (SRS with precomputed window tables T_w[i] = 2^(c w) P_i, built once at upload — HBM capacity is
//                 spent to remove work): every (scalar, window) entry adds +-T_w[i] into ONE set of 2^(c-1) buckets;
//                 W n mixed adds, one shuffle-based bucket reduction, no Horner.
//   generic mode (caller-provided bases, e.g. g1_lincomb; tiny or huge SRS): W bucket sets, per-window reduction,
//                 Horner over the W window sums on the host.
#include "engine.h"
#include <new>
#include "msm_kernels.h"
#include "host_curve.h"
#include "host_pairing.h"      // fr_wire_to_canonical (KZG_DEBUG_SORT)

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

namespace kzg {

constexpr uint32_t MSM_BATCH_POLYS_MAX = 1024;  // polynomials of one batched table-mode launch (64 buckets each: 2^16 buckets)
constexpr uint32_t MSM_MAX_OUT = 16384;        // XYZZ values one launch may hand to the host epilogue (generic mode: W * batch window sums)
constexpr size_t SORT1_MAX_LDS = 131072;       // single-pass sort: one LDS counter per bucket (<= 2^15 buckets)

void MsmWorkspace::release() {
    DeviceBuffer* all[] = {&scalars, &bases, &bases_wire, &digits, &sorted, &count, &blockbase, &sort_tmp, &sort_key, &sort_small, &offs, &block_sums,
                           &head, &cont, &blob, &bucket, &chunkS, &chunkTmp, &chunkA, &out_wire};
    for (auto* b : all) b->release();
    if (pinned_out) { (void)hipHostFree(pinned_out); pinned_out = nullptr; }
    if (ev_ready) { for (auto& e : ev) (void)hipEventDestroy(e); ev_ready = false; }
    if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
    if (ev_sorted) { (void)hipEventDestroy(ev_sorted); ev_sorted = nullptr; }
}

static int ilog2_floor(size_t n) { int k = 0; while ((n >> (k + 1)) != 0) ++k; return k; }

struct Plan {
    uint32_t n;          // pairs per MSM in this launch
    uint32_t batch;      // independent MSMs of n pairs each (generic mode only; 1 otherwise)
    bool tables;         // table mode
    bool naf;            // table mode over the per-bit tables: width-(c + 1) NAF digits (msm_kernels.h), W = most entries per scalar
    uint32_t polys;      // BATCHED table mode (NAF, c = 7): the n scalars are `polys` polynomials of n / polys coefficients over the same bases,
                         // 64 buckets each: B = 64 polys buckets, one group of the first reduction level per polynomial (0: one MSM)
    int c, W;
    uint32_t B;          // buckets per set
    uint32_t sets;       // bucket sets (1 in table mode, W otherwise)
    uint32_t G;          // sets * B
    uint32_t nl;         // lanes of the accumulate kernel (a multiple of 256); each adds ceil(E / nl) sorted entries
    uint32_t set_len;    // digit entries per set
    uint32_t tile_len, tiles_per_set, tiles;
    bool bitsum;         // tiny MSM over the per-bit tables as a plain sum: one result point (k_bitsum_level1 / 2)
    bool fused;          // sparse table-mode MSM: the first reduction level adds the sorted entries itself; its group g holds the buckets gp * G1 + g (set by msm_enqueue)
    bool quad;           // reduction levels on lane quads (curve_quad.h): no other MSM in flight when this one was planned
    bool alone = false;  // planned with no other MSM of this context in flight
    bool sort2;          // two-level sort (table mode, index fits 24 bits)
    bool sort_small;     // global-atomic sort (few entries)
    uint32_t Hb, tile1, tiles1, tiles2cap;
    uint32_t T, m;       // generic-mode reduction: chunks per window, buckets per chunk
    // table mode: a sorted entry holds window * idx_stride + i.  idx_stride = the table stride, or (compact form, SRS of more than
    // 2^20 points) the next power of two >= n, 2^idx_log, with the accumulate kernel adding window * stride_adj
    uint32_t idx_stride, idx_log, stride_adj;
};

// Window bits of the generic mode for `batch` MSMs of n pairs: W * batch window sums leave the device (<= MSM_MAX_OUT).
static int generic_window(size_t n, uint32_t batch) {
    int c = std::min(14, std::max(4, ilog2_floor(n) - 6));
    while (c < 16 && (size_t)((255 + c - 1) / c) * batch > MSM_MAX_OUT) ++c;
    return c;
}

// Batched table mode: bucket bits per polynomial by its length (width c + 1 digits over the per-bit tables).  Short polynomials: 64
// buckets, one group of the first reduction level each; from 2^13 coefficients whole units of 4 096 buckets, reduced like a single MSM.
static int batch_bucket_bits(size_t poly_len) {
    if (poly_len < ((size_t)1 << 13)) return 7;
    if (poly_len < ((size_t)1 << 15)) return 13;
    if (poly_len < ((size_t)1 << 18)) return 15;
    return 16;
}
static Plan make_plan(const kzg_ctx* ctx, size_t n, const MsmBases& bases, uint32_t batch, uint32_t polys = 0) {
    Plan p;
    p.n = (uint32_t)n;
    p.batch = batch;
    p.tables = bases.table_stride != 0;
    p.naf = p.tables && bases.naf;
    p.polys = p.naf ? polys : 0;
    p.fused = false;
    p.bitsum = false;
    {
        // lane quads for the two reduction levels when this MSM runs alone (0.7 of the pair form's dependent instructions, twice its lanes);
        // with another MSM in flight the SIMDs are shared and the pair form's fewer instructions count (kzg_ctx_set_reduction_lanes forces either).
        bool other_in_flight = false;
        for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl) other_in_flight |= ctx->slot_pending[sl] != nullptr;
        p.quad = p.tables && !other_in_flight;
        if (ctx->reduction_lanes) p.quad = p.tables && ctx->reduction_lanes == 4;
        p.alone = !other_in_flight;
    }
    int c;
    if (p.polys) {
        c = batch_bucket_bits(n / p.polys);                // 7: 64 buckets per polynomial (k_batch_finish); 13 / 15 / 16: whole units of 4 096 buckets (second level + host epilogue per polynomial)
    } else if (p.tables) {
        c = bases.c;
        // NAF mode, 2^18 .. 2^19 - 1 pairs, nothing else in flight (the reference's bench_kzg_commit_8mb shape): 2^14 buckets instead of 2^15 --
        // alone, the reductions cost their latency, not their instructions: 0.551 -> 0.525 ms at 2^18 (with other MSMs in flight 16 stays
        // ahead, engine.h srs_naf_c; tools/archive/sweep_naf_c_alone.py).
        if (p.naf && c == 16 && p.alone && n < ((size_t)1 << 19)) c = 15;
    } else {
        c = ctx->msm_c_override;
        if (c == 0) c = generic_window(n, batch);
        c = std::min(16, std::max(2, c));
    }
    p.c = c;
    p.W = p.naf ? naf_max_digits(c + 1) : (255 + c - 1) / c;
    p.B = p.polys ? (c == 7 ? 64u * ((p.polys + 1u) & ~1u) : p.polys << (c - 1)) : 1u << (c - 1);      // (a multiple of 128: whole coarse bins)
    p.sets = p.tables ? 1u : (uint32_t)p.W * batch;
    p.G = p.sets * p.B;
    const size_t entries_cap = (size_t)p.W * n * batch;                  // buffer sizes
    // NAF: the entry count is only known on the device (offs[G]); 254 / (w + 1) per scalar on average sizes the accumulate grid
    const size_t entries = p.naf ? std::max<size_t>(1, (size_t)((double)n * 254.0 / (double)(c + 2))) : entries_cap;
    {
        // Lanes of the accumulate kernel.  Large MSMs: one full round of resident waves (ctx->acc_wave_slots = 3 per SIMD), every
        // lane with the same trip count.  Small MSMs: at least Lmin entries per lane -- short trips keep them from serialising
        // ~100 dependent mixed adds (10 us each) in a handful of waves; from 2^21 entries on, below 24 entries per lane the
        // folding of the lane partials costs more than the extra waves buy (measured in round 1: 15 -> 0.655 ms, 24 -> 0.582 ms).
        const int L = ctx->msm_seg_override;
        size_t lanes;
        if (L > 0) {
            lanes = (entries + (size_t)L - 1) / (size_t)L;                 // forced trip count (tests, sweeps): no cap
        } else {
            const size_t lmin = entries >= ((size_t)1 << 21) ? 24 : 4;
            // Round 3: TWO waves per SIMD (of the three the 168-VGPR kernel could hold) unless this is a large MSM running alone.
            // A grid that fills all three slots leaves no registers for any other kernel on the chip, so the sort and the reductions
            // of the other MSM in flight only ran in the tail of this kernel; with a third of the slots free they run beside it:
            // pipelined step 1.183-1.192 -> 1.160-1.166 ms (same box, 3072 / 2048 wave slots; 2560 = 2.5 waves per SIMD: 1.17-1.18),
            // and below 2^19 pairs fewer lanes also mean fewer partial sums for the first reduction level (2^16: 0.464 -> 0.403 ms,
            // 2^17: 0.534 -> 0.486).  Alone, a 2^19 / 2^20-pair MSM is 2-3 % faster on three (0.929 / 1.474 against 0.961 / 1.496 ms).
            size_t slots = ctx->acc_wave_slots;
            {
                bool other_in_flight = false;
                for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl) other_in_flight |= ctx->slot_pending[sl] != nullptr;
                if (other_in_flight || entries < ((size_t)1 << 23)) slots = slots / 3 * 2;
                // below 2^22 entries (2^15 .. 2^17 pairs; 2^18 is even) ONE wave per SIMD: the kernel is not throughput bound there (the same
                // 0.20 ms at 2^17 pairs with 65 536 lanes of 30 entries as with 131 072 of 15), half the lanes leave half the partial sums
                // to the first reduction level (0.097 -> 0.084 ms) and room for the other MSMs in flight: three in flight 2^15 0.154 ->
                // 0.136, 2^16 0.191 -> 0.160, 2^17 0.245 -> 0.211 ms per MSM (wave-slot sweep, profiles/r03_naf.md)
                if (entries < ((size_t)1 << 22)) slots = ctx->acc_wave_slots / 3;
            }
            lanes = std::min<size_t>(slots * 64, (entries + lmin - 1) / lmin);
        }
        lanes = std::max<size_t>(256, (lanes + 255) / 256 * 256);
        p.nl = (uint32_t)std::min<size_t>(lanes, (size_t)1 << 24);
    }
    p.set_len = (uint32_t)(p.tables ? entries_cap : n);
    p.idx_stride = bases.table_stride; p.idx_log = 31; p.stride_adj = 0;
    if (p.tables && !p.naf && (size_t)p.W * bases.table_stride > ((size_t)1 << SORT2_IDX_BITS)) {
        int lg = ilog2_floor(n);
        if (((size_t)1 << lg) < n) ++lg;
        if (((size_t)p.W << lg) <= ((size_t)1 << SORT2_IDX_BITS) && ((size_t)1 << lg) <= bases.table_stride) {
            p.idx_log = (uint32_t)lg; p.idx_stride = 1u << lg; p.stride_adj = bases.table_stride - p.idx_stride;
        }
    }
    // single-pass sort tiles: large against the bucket count (one contiguous flush of B counters per tile), and not too many
    size_t tile = std::max<size_t>(4096, 2 * (size_t)p.B);
    while (tile < p.set_len && ((size_t)p.set_len + tile - 1) / tile * p.sets > 1024) tile *= 2;   // (many small sets: one tile per set)
    p.tile_len = (uint32_t)tile;
    p.tiles_per_set = (uint32_t)(((size_t)p.set_len + tile - 1) / tile);
    p.tiles = p.tiles_per_set * p.sets;
    {
        const bool lds_fits = (size_t)p.B * 4 <= SORT1_MAX_LDS;            // single-pass sort: one LDS counter per bucket
        p.sort_small = !p.naf && entries < ((size_t)1 << 18);
        const bool can2 = p.tables && (p.c - 1 > SORT2_LO_BITS || p.polys) && (p.B >> SORT2_LO_BITS) <= SORT2_MAX_BINS &&
                          (p.naf ? (size_t)NAF_POSITIONS * p.idx_stride < ((size_t)1 << 31)
                                 : (size_t)p.W * p.idx_stride <= ((size_t)1 << SORT2_IDX_BITS));
        // from 2^18 entries (was 2^23): the scalar-tile pass 1 and the per-bin pass 2 win from the first size the single-pass sort is not "small" for
        p.sort2 = !p.sort_small && can2 && (entries >= ((size_t)1 << 18) || !lds_fits || p.naf);
        if (!p.sort2 && !lds_fits) p.sort_small = true;                    // (slow but correct: a forced odd configuration)
        p.Hb = p.sort2 ? (p.B >> SORT2_LO_BITS) : 0;
        p.tile1 = n >= ((size_t)1 << 19) ? 2048 : 1024;                    // SCALARS per pass-1 tile (W entries each)
        if (p.naf) {                                                       // the recoding is a long dependent chain per scalar: more, smaller tiles
            p.tile1 = 512;
            if (p.polys) p.tile1 = 2048;                                    // (twice the entries per scalar: fewer, larger tiles)
        }
        p.tiles1 = (uint32_t)((n + p.tile1 - 1) / p.tile1);
        p.tiles2cap = (uint32_t)(entries_cap / SORT2_CHUNK + p.Hb + 1);
    }
    p.T = std::min<uint32_t>(p.B, RED_T);
    p.m = p.B / p.T;
    return p;
}

// Largest number of pairs one launch takes: W * n must fit the 32-bit positions of the sort.
static const size_t MSM_MAX_LAUNCH = (size_t)1 << 24;

// What msm_enqueue leaves in flight on its stream; msm_finish waits for it and runs the host epilogue.
struct Pending {
    Plan p;
    uint32_t n_out = 0;
    uint32_t batch = 1;
    size_t n = 0;
    uint32_t out_off = 0;    // first point of this launch's results in the pinned result buffer
    bool profiled = true;    // the workspace's phase events belong to this launch (the last one enqueued)
};
// One asynchronous MSM = up to MSM_MAX_PARTS launches back to back on the slot's stream, sharing its workspace (stream order keeps
// them apart); each copies its O(200) result points to its own MSM_PART_OUT-point window of the pinned buffer.
constexpr uint32_t MSM_MAX_PARTS = 64;       // round 4: 64 launches (2^26 pairs over a table-mode SRS of more than 2^20 points); 16 before
constexpr uint32_t MSM_PART_OUT = MSM_MAX_OUT / MSM_MAX_PARTS;
struct MsmPending {
    Pending part[MSM_MAX_PARTS];
    uint32_t n_parts = 0;
};

// Enqueue every kernel of one MSM (or batch) plus the D2H copy of its O(100) result points on `st`, using `ws`.
static int32_t msm_enqueue(kzg_ctx* ctx, MsmWorkspace& ws, hipStream_t st, const MsmBases& bases, const uint4* d_scalars, size_t n,
                           uint32_t batch, Pending* pend, uint32_t out_off = 0, uint32_t out_cap = MSM_MAX_OUT, uint32_t polys = 0,
                           const PolyPtrs* poly_ptrs = nullptr) {
    if (batch == 0 || (batch > 1 && bases.table_stride != 0)) return KZG_ERR_INVALID_ARG;
    if (polys && (!bases.naf || n % polys != 0 || polys > MSM_BATCH_POLYS_MAX)) return KZG_ERR_INVALID_ARG;
    if (bases.bitsum && batch == 1 && !polys && n <= BITSUM_MAX_N) {
        // tiny MSM: two launches, one result point (msm_kernels.h section 6e)
        RoctxRange range_bs("kzg:msm:bit sums");
        Plan p{};
        p.n = (uint32_t)n; p.batch = 1; p.tables = true; p.bitsum = true; p.B = 64; p.G = 64; p.W = 255; p.c = 7;
        const int chunk = n <= 512 ? 8 : n <= 2048 ? 16 : 32;                     // positions per quad: one wave per SIMD at 512 / 1 024 scalars (the six-step tree of every workgroup is what the launch costs: 2^9 30 us with 8, 38 with 4; 2^11 63 with 16, 77 with 8)
        const uint32_t n_wg = (uint32_t)((n * (size_t)(256 / chunk) + 63) / 64);
        KZG_HIP_TRY(ctx, ws.chunkS.reserve((size_t)n_wg * 36 * 4));
        if (!ws.pinned_out) {
            KZG_HIP_TRY(ctx, hipHostMalloc(&ws.pinned_out, (size_t)MSM_MAX_OUT * 32 * 4 + MSM_MAX_PARTS * 4, hipHostMallocDefault));
            KZG_HIP_TRY(ctx, hipHostGetDevicePointer(&ws.pinned_out_dev, ws.pinned_out, 0));
        }
        const uint32_t n_res = (n_wg + 63) / 64;                 // <= 8 points for the host to add
        if (out_cap < n_res || out_off + n_res > MSM_MAX_OUT) return KZG_ERR_INVALID_ARG;
        uint32_t* d_out = reinterpret_cast<uint32_t*>(ws.pinned_out_dev) + (size_t)out_off * 32;
        const bool prof = ctx->profiling;
        if (prof && !ws.ev_ready) {
            for (auto& e : ws.ev) KZG_HIP_TRY(ctx, hipEventCreate(&e));
            ws.ev_ready = true;
        }
        if (prof) {
            reinterpret_cast<uint32_t*>(static_cast<char*>(ws.pinned_out) + (size_t)MSM_MAX_OUT * 128)[out_off / MSM_PART_OUT] = 0;
            for (int i = 0; i <= 5; ++i) KZG_HIP_TRY(ctx, hipEventRecord(ws.ev[i], st));
        }
        if (chunk == 8)
            hipLaunchKernelGGL(k_bitsum_level1<8>, dim3(n_wg), dim3(256), 0, st, bases.points, bases.table_stride, d_scalars, (uint32_t)n, ws.chunkS.as<int32_t>());
        else if (chunk == 16)
            hipLaunchKernelGGL(k_bitsum_level1<16>, dim3(n_wg), dim3(256), 0, st, bases.points, bases.table_stride, d_scalars, (uint32_t)n, ws.chunkS.as<int32_t>());
        else
            hipLaunchKernelGGL(k_bitsum_level1<32>, dim3(n_wg), dim3(256), 0, st, bases.points, bases.table_stride, d_scalars, (uint32_t)n, ws.chunkS.as<int32_t>());
        if (prof) KZG_HIP_TRY(ctx, hipEventRecord(ws.ev[6], st));
        hipLaunchKernelGGL(k_bitsum_level2, dim3(n_res), dim3(256), 0, st, ws.chunkS.as<int32_t>(), n_wg, d_out);
        if (prof) KZG_HIP_TRY(ctx, hipEventRecord(ws.ev[7], st));
        KZG_HIP_TRY(ctx, hipGetLastError());
        if (!ws.ev_done) KZG_HIP_TRY(ctx, hipEventCreateWithFlags(&ws.ev_done, hipEventDisableTiming));
        KZG_HIP_TRY(ctx, hipEventRecord(ws.ev_done, st));
        pend->p = p;
        pend->n_out = n_res;
        pend->batch = 1;
        pend->n = n;
        pend->out_off = out_off;
        pend->profiled = true;
        return KZG_OK;
    }
    Plan p = make_plan(ctx, n, bases, batch, polys);
    if (p.polys && p.B > 65536) return KZG_ERR_INVALID_ARG;
    const int ND = p.c + 1 >= 16 ? NAF_DIGITS : 32;        // digit words per scalar (width >= 16: at most 16 digits)
    const size_t entries = (size_t)p.W * n * batch;
    const uint32_t n_windows = (uint32_t)p.W * batch;          // window sums produced in generic mode
    if (!p.tables && n_windows > MSM_MAX_OUT) return KZG_ERR_INVALID_ARG;
    const uint32_t n_chunks = n_windows * p.T;
    const uint32_t nb = (p.G + SCAN_TILE - 1) / SCAN_TILE;
    if (p.G > SCAN1_MAX && nb > SCAN_TILE) return KZG_ERR_INVALID_ARG;
    const uint32_t G1 = p.B / 64, G1p = (G1 + 63) / 64;
    if (p.tables && G1 > 1 && 13 * G1p > MSM_MAX_OUT) return KZG_ERR_INVALID_ARG;

    if (!p.sort2) KZG_HIP_TRY(ctx, ws.digits.reserve(entries * 4));
    KZG_HIP_TRY(ctx, ws.sorted.reserve(entries * 4));
    KZG_HIP_TRY(ctx, ws.count.reserve((size_t)p.G * 4 + 16));
    if (p.sort2) {
        KZG_HIP_TRY(ctx, ws.sort_tmp.reserve(entries * 4));
        if (p.naf) {
            KZG_HIP_TRY(ctx, ws.sort_key.reserve(entries + 16));
            KZG_HIP_TRY(ctx, ws.digits.reserve((size_t)n * ND * 4));
        }
        KZG_HIP_TRY(ctx, ws.sort_small.reserve(((size_t)3 * (p.Hb + 1) + p.tiles2cap) * 4 + 64));
        KZG_HIP_TRY(ctx, ws.blockbase.reserve(std::max((size_t)p.tiles1 * p.Hb, (size_t)p.tiles2cap * SORT2_LO) * 4));
    } else if (p.sort_small) {
        KZG_HIP_TRY(ctx, ws.blockbase.reserve((size_t)p.G * 4));
    } else {
        KZG_HIP_TRY(ctx, ws.blockbase.reserve((size_t)p.tiles * p.B * 4));
    }
    KZG_HIP_TRY(ctx, ws.offs.reserve(((size_t)p.G + 1) * 4 + 16));
    KZG_HIP_TRY(ctx, ws.block_sums.reserve((size_t)SCAN_TILE * 4));
    KZG_HIP_TRY(ctx, ws.head.reserve((size_t)p.G * 36 * 4));
#ifdef KZG_ACC_STAMPS
    KZG_HIP_TRY(ctx, ws.cont.reserve((size_t)p.nl * 36 * 4 + (size_t)(p.nl / 64) * 64));
#else
    KZG_HIP_TRY(ctx, ws.cont.reserve((size_t)p.nl * 36 * 4));
#endif
    if (p.tables) {
        KZG_HIP_TRY(ctx, ws.chunkS.reserve((size_t)7 * G1 * 36 * 4));          // X1
    } else {
        KZG_HIP_TRY(ctx, ws.bucket.reserve((size_t)p.G * 36 * 4));
        KZG_HIP_TRY(ctx, ws.chunkS.reserve((size_t)n_chunks * 36 * 4));
        KZG_HIP_TRY(ctx, ws.chunkTmp.reserve((size_t)n_chunks * 36 * 4));
        KZG_HIP_TRY(ctx, ws.chunkA.reserve((size_t)n_chunks * 36 * 4));
    }
    if (!ws.pinned_out) {
        KZG_HIP_TRY(ctx, hipHostMalloc(&ws.pinned_out, (size_t)MSM_MAX_OUT * 32 * 4 + MSM_MAX_PARTS * 4, hipHostMallocDefault));   // + entry counts of profiled launches
        KZG_HIP_TRY(ctx, hipHostGetDevicePointer(&ws.pinned_out_dev, ws.pinned_out, 0));
    }
    // The last kernel of the sequence stores the O(200) result points straight into the pinned host buffer (coherent host memory, read
    // after the event behind that kernel).  A device-to-host copy of them cost ~10 us per MSM -- and above ~16 KiB (208 points at 2^16
    // buckets: every batched launch) hipMemcpyAsync takes the SDMA path, whose set-up after a device-wide synchronisation blocked the
    // enqueueing thread for 5.6-7 ms (gone with HSA_ENABLE_SDMA=0).
    uint32_t* d_out = reinterpret_cast<uint32_t*>(ws.pinned_out_dev) + (size_t)out_off * 32;
    if (!ctx->lds_attr_set) {
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SORT1_MAX_LDS));
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SORT1_MAX_LDS));
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort2_scatter1_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)((3 * SORT2_MAX_BINS + SORT2_P1_THREADS * 31) * 4)));
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort2_scatter1_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)((3 * SORT2_MAX_BINS + SORT2_P1_THREADS * 31) * 4 + SORT2_P1_THREADS * 31 * 2)));
        KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort2_scatter1_lds<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)((3 * SORT2_MAX_BINS + SORT2_P1_THREADS * 32) * 4 + SORT2_P1_THREADS * 32 * 2)));
        ctx->lds_attr_set = true;
    }

    const bool prof = ctx->profiling;
    if (prof && !ws.ev_ready) {
        for (auto& e : ws.ev) KZG_HIP_TRY(ctx, hipEventCreate(&e));
        ws.ev_ready = true;
    }
#define KZG_MARK(i) do { if (prof) KZG_HIP_TRY(ctx, hipEventRecord(ws.ev[i], st)); } while (0)

    // Staggered start.  Two MSMs enqueued back to back on two streams (the fill of a pipeline) run their phases in lock-step: both sort
    // (memory-bound) and then both accumulate (VALU-bound), instead of one sorting beside the other's accumulate as in steady state,
    // where MSM k + 2 is only enqueued once MSM k is done.  So this launch starts behind the SORT of the previous launch of the context
    // when that went to another stream -- a no-op in steady state (that sort finished long ago), the natural stagger at the fill:
    // 20-step regions (profiles/r04_ab_msm_stagger.txt): 2^17-pair steps, four per launch, 0.188 -> 0.183 ms per step; 2^18 / 2^19, two per launch,
    // -0.7 %; long regions unchanged.  Only GROUPED launches wait (the sharded streams): a single 2^20-pair MSM planned alone lost 0.7 %
    // (1.115 -> 1.122: its successor's sort starts 0.2 ms later and the first MSM runs on three wave slots either way).
    const bool stagger = p.tables && p.polys >= 2;
    if (stagger && ctx->last_sorted && ctx->last_sorted_stream != st) KZG_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->last_sorted, 0));

    uint32_t* d_offs = ws.offs.as<uint32_t>();
    auto scan_counts = [&]() {
        if (p.G <= SCAN1_MAX) {
            hipLaunchKernelGGL(k_scan_counts_1wg<false>, dim3(1), dim3(SCAN1_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, d_offs, (uint32_t*)nullptr);
        } else {
            hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, ws.block_sums.as<uint32_t>());
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, st, ws.block_sums.as<uint32_t>(), nb);
            hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, ws.block_sums.as<uint32_t>(), d_offs);
        }
    };

    KZG_MARK(0);
    RoctxPhases phases;
    phases.begin("kzg:msm:sort");
    // what an earlier small sort left of `count` (see k_scan_counts_1wg<true>); every other path writes the counters as it likes
    const uint32_t clean_g = ws.count_zero_ptr == ws.count.p ? ws.count_zero_g : 0;
    ws.count_zero_g = 0;
    const bool lean_sort = !p.sort2 && p.sort_small && p.tables && batch == 1 && p.G == p.B && p.G <= SCAN1_MAX;
    const uint32_t n_total = p.n * batch;
    const uint32_t gn = (n_total + 255) / 256;
    const size_t lds_bytes = (size_t)p.B * 4;
    if (p.sort2) {
        // two-level sort straight from the scalars (no digit array)
        uint32_t* small = ws.sort_small.as<uint32_t>();
        uint32_t* ccount = small;                       // Hb
        uint32_t* cstart = small + (p.Hb + 1);          // Hb + 1
        uint32_t* tstart = small + 2 * (p.Hb + 1);      // Hb + 1
        uint32_t* tile_bin = small + 3 * (p.Hb + 1);    // tiles2cap
        uint32_t* bin_cap = tile_bin + p.tiles2cap;     // 1: the LARGE-bin threshold of this launch (k_sort2_scan)
        KZG_HIP_TRY(ctx, hipMemsetAsync(ccount, 0, (size_t)p.Hb * 4, st));
        uint8_t* tmpk = p.naf ? ws.sort_key.as<uint8_t>() : nullptr;
        const uint32_t poly_len = p.polys ? p.n / p.polys : 0u;
        PolyPtrs ptrs{};
        if (poly_ptrs) ptrs = *poly_ptrs;
        if (p.naf && ND == 32)
            hipLaunchKernelGGL(k_naf_digits<32>, dim3(p.tiles1), dim3(256), ((size_t)p.Hb + (11 + 32) * 256) * 4, st, d_scalars, p.n, p.c, p.tile1, p.Hb, ccount,
                               ws.blockbase.as<uint32_t>(), ws.digits.as<uint4>(), poly_len, ptrs);
        else if (p.naf)
            hipLaunchKernelGGL(k_naf_digits<NAF_DIGITS>, dim3(p.tiles1), dim3(256), ((size_t)p.Hb + (11 + NAF_DIGITS) * 256) * 4, st, d_scalars, p.n, p.c, p.tile1, p.Hb, ccount,
                               ws.blockbase.as<uint32_t>(), ws.digits.as<uint4>(), poly_len, ptrs);
        else
        hipLaunchKernelGGL(k_sort2_scalars<false>, dim3(p.tiles1), dim3(256), (size_t)p.Hb * 4, st, d_scalars, p.n, p.c, p.W, p.tile1, p.Hb, ccount,
                           ws.blockbase.as<uint32_t>(), (const uint32_t*)nullptr, p.idx_stride, (uint32_t*)nullptr);
        KZG_MARK(1);
        hipLaunchKernelGGL(k_sort2_scan, dim3(1), dim3(512), 0, st, ccount, p.Hb, cstart, tstart, tile_bin, ws.count.as<uint32_t>(), bin_cap);
        if (p.naf) {
            const size_t lds1 = ((size_t)3 * p.Hb + (size_t)SORT2_P1_THREADS * p.W) * 4 + (size_t)SORT2_P1_THREADS * p.W * 2;
            if (ND == 32)
                hipLaunchKernelGGL((k_sort2_scatter1_lds<true, 32>), dim3(p.tiles1), dim3(SORT2_P1_THREADS), lds1, st, ws.digits.as<uint4>(), p.n, p.c, p.W, p.tile1, p.Hb,
                                   ws.blockbase.as<uint32_t>(), (const uint32_t*)cstart, p.idx_stride, ws.sort_tmp.as<uint32_t>(), tmpk, poly_len);
            else
            hipLaunchKernelGGL(k_sort2_scatter1_lds<true>, dim3(p.tiles1), dim3(SORT2_P1_THREADS), lds1, st, ws.digits.as<uint4>(), p.n, p.c, p.W, p.tile1, p.Hb,
                               ws.blockbase.as<uint32_t>(), (const uint32_t*)cstart, p.idx_stride, ws.sort_tmp.as<uint32_t>(), tmpk, poly_len);
        } else if (p.W > 31) {                              // (more windows than the LDS staging holds: pass 1 scatters directly)
            hipLaunchKernelGGL(k_sort2_scalars<true>, dim3(p.tiles1), dim3(256), (size_t)p.Hb * 4, st, d_scalars, p.n, p.c, p.W, p.tile1, p.Hb, ccount,
                               ws.blockbase.as<uint32_t>(), (const uint32_t*)cstart, p.idx_stride, ws.sort_tmp.as<uint32_t>());
        } else {
            const size_t lds1 = ((size_t)3 * p.Hb + (size_t)SORT2_P1_THREADS * p.W) * 4;
            hipLaunchKernelGGL(k_sort2_scatter1_lds<false>, dim3(p.tiles1), dim3(SORT2_P1_THREADS), lds1, st, d_scalars, p.n, p.c, p.W, p.tile1, p.Hb,
                               ws.blockbase.as<uint32_t>(), (const uint32_t*)cstart, p.idx_stride, ws.sort_tmp.as<uint32_t>(), (uint8_t*)nullptr);
        }
        KZG_MARK(2);
        if (p.naf) {
            hipLaunchKernelGGL(k_sort2_hist2<true>, dim3(p.tiles2cap), dim3(256), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, tstart, tile_bin, p.Hb,
                               ws.count.as<uint32_t>(), ws.blockbase.as<uint32_t>(), (const uint8_t*)tmpk);
            hipLaunchKernelGGL(k_sort2_bin<true>, dim3(p.Hb), dim3(SORT2_BIN_THREADS), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, p.Hb, ws.count.as<uint32_t>(),
                               d_offs, ws.sorted.as<uint32_t>(), (const uint32_t*)bin_cap, (const uint8_t*)tmpk);
            hipLaunchKernelGGL(k_sort2_scatter2<true>, dim3(p.tiles2cap), dim3(256), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, tstart, tile_bin, p.Hb,
                               d_offs, ws.blockbase.as<uint32_t>(), ws.sorted.as<uint32_t>(), (const uint8_t*)tmpk);
        } else {
        hipLaunchKernelGGL(k_sort2_hist2<false>, dim3(p.tiles2cap), dim3(256), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, tstart, tile_bin, p.Hb,
                           ws.count.as<uint32_t>(), ws.blockbase.as<uint32_t>(), (const uint8_t*)nullptr);
        hipLaunchKernelGGL(k_sort2_bin<false>, dim3(p.Hb), dim3(SORT2_BIN_THREADS), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, p.Hb, ws.count.as<uint32_t>(),
                           d_offs, ws.sorted.as<uint32_t>(), (const uint32_t*)bin_cap, (const uint8_t*)nullptr);
        hipLaunchKernelGGL(k_sort2_scatter2<false>, dim3(p.tiles2cap), dim3(256), 0, st, ws.sort_tmp.as<uint32_t>(), cstart, tstart, tile_bin, p.Hb,
                           d_offs, ws.blockbase.as<uint32_t>(), ws.sorted.as<uint32_t>(), (const uint8_t*)nullptr);
        }
        KZG_MARK(3);
    } else if (lean_sort) {
        // small table-mode MSM (one bucket set): digits + histogram in one pass, the scan leaves the scatter's cursors and clears the counters
        // it read -- three launches instead of four launches and two memsets (each ~3-5 us of a ~100 us commitment)
        if (clean_g < p.G) KZG_HIP_TRY(ctx, hipMemsetAsync(ws.count.p, 0, (size_t)p.G * 4, st));
        hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(256), 0, st, d_scalars, n_total, p.n, p.c, p.W, ws.digits.as<uint32_t>(), ws.count.as<uint32_t>());
        KZG_MARK(1);
        hipLaunchKernelGGL(k_scan_counts_1wg<true>, dim3(1), dim3(SCAN1_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, d_offs, ws.blockbase.as<uint32_t>());
        KZG_MARK(2);
        const uint32_t ge = (uint32_t)((entries + 255) / 256);
        hipLaunchKernelGGL(k_sort_small_scatter, dim3(ge), dim3(256), 0, st, ws.digits.as<uint32_t>(), (uint32_t)entries, p.n, p.set_len, p.B,
                           d_offs, ws.blockbase.as<uint32_t>(), p.idx_stride, (uint32_t)p.W, ws.sorted.as<uint32_t>(), 1);
        KZG_MARK(3);
        ws.count_zero_ptr = ws.count.p;
        ws.count_zero_g = p.G;
    } else {
        KZG_HIP_TRY(ctx, hipMemsetAsync(ws.count.p, 0, (size_t)p.G * 4, st));
        hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(256), 0, st, d_scalars, n_total, p.n, p.c, p.W, ws.digits.as<uint32_t>(), (uint32_t*)nullptr);
        KZG_MARK(1);
        if (p.sort_small) {
            const uint32_t ge = (uint32_t)((entries + 255) / 256);
            KZG_HIP_TRY(ctx, hipMemsetAsync(ws.blockbase.p, 0, (size_t)p.G * 4, st));
            hipLaunchKernelGGL(k_sort_small_hist, dim3(ge), dim3(256), 0, st, ws.digits.as<uint32_t>(), (uint32_t)entries, p.set_len, p.B,
                               ws.count.as<uint32_t>());
            scan_counts();
            KZG_MARK(2);
            hipLaunchKernelGGL(k_sort_small_scatter, dim3(ge), dim3(256), 0, st, ws.digits.as<uint32_t>(), (uint32_t)entries, p.n, p.set_len, p.B,
                               d_offs, ws.blockbase.as<uint32_t>(), p.idx_stride, (uint32_t)p.W, ws.sorted.as<uint32_t>());
        } else {
            hipLaunchKernelGGL(k_sort_hist, dim3(p.tiles), dim3(256), lds_bytes, st, ws.digits.as<uint32_t>(), p.set_len, p.tile_len,
                               p.tiles_per_set, p.B, ws.count.as<uint32_t>(), ws.blockbase.as<uint32_t>());
            scan_counts();
            KZG_MARK(2);
            hipLaunchKernelGGL(k_sort_scatter, dim3(p.tiles), dim3(256), lds_bytes, st, ws.digits.as<uint32_t>(), p.n, p.set_len, p.tile_len,
                               p.tiles_per_set, p.B, d_offs, ws.blockbase.as<uint32_t>(), p.idx_stride,
                               (uint32_t)p.W, ws.sorted.as<uint32_t>());
        }
        KZG_MARK(3);
    }
    KZG_MARK(4);
    if (p.tables) {                                         // (recorded by every table-mode launch: the NEXT launch decides whether it waits)
        if (!ws.ev_sorted) KZG_HIP_TRY(ctx, hipEventCreateWithFlags(&ws.ev_sorted, hipEventDisableTiming));
        KZG_HIP_TRY(ctx, hipEventRecord(ws.ev_sorted, st));
        ctx->last_sorted = ws.ev_sorted;
        ctx->last_sorted_stream = st;
    }
    phases.begin("kzg:msm:accumulate");
    // the two reduction levels run on lane pairs (curve_pair.h; one 128-thread workgroup per 64 buckets, then per two groups of 64 sums) or lane quads
    // sparse table-mode MSMs (at most 2.5 entries per bucket on average: commitments of <= 2^11 coefficients on
    // the c = 15 tables): the first reduction level adds the entries itself (k_msm_bucket_bits1p_fused), there is no accumulate kernel and
    // there are no partial sums.  Measured (tools/phases_small.py, same box, device time of one commitment): 2^8 170 -> 125 us, 2^9 163 -> 128,
    // 2^10 165 -> 152, 2^11 199 -> 195; at 2^12 (4.25 per bucket) 206 -> 261: a wave waits for its fullest bucket, the equal split does not.
    const bool fused = p.tables && !p.naf && (double)entries <= 2.5 * (double)p.B;
    p.fused = fused;
    // (64- and 128-thread workgroups measured the same as 256)
    if (!fused)
        hipLaunchKernelGGL(k_msm_accumulate, dim3(p.nl / 256), dim3(256), 0, st, bases.points, ws.sorted.as<uint32_t>(), d_offs, p.G,
                           ws.head.as<int32_t>(), (size_t)p.G, ws.cont.as<int32_t>(), (size_t)p.nl, p.idx_log, p.stride_adj);
    KZG_MARK(5);
    phases.begin("kzg:msm:bucket reduction");
    uint32_t n_out;                       // wire XYZZ values copied to the host
    if (p.tables) {
        if (fused && p.quad)
            hipLaunchKernelGGL(k_msm_bucket_bits1q_fused, dim3(G1), dim3(256), 0, st, bases.points, ws.sorted.as<uint32_t>(), d_offs, p.B, p.idx_log,
                               p.stride_adj, G1, ws.chunkS.as<int32_t>(), (size_t)7 * G1, d_out);
        else if (fused)
            hipLaunchKernelGGL(k_msm_bucket_bits1p_fused, dim3(G1), dim3(128), 0, st, bases.points, ws.sorted.as<uint32_t>(), d_offs, p.B, p.idx_log,
                               p.stride_adj, G1, ws.chunkS.as<int32_t>(), (size_t)7 * G1, d_out);
        else if (p.quad && (G1 <= 512 || ctx->reduction_lanes == 4))
            // (at 2^16 buckets the level is 4 096 quad waves of ~11 000 instructions: throughput bound, 0.121 against 0.114 ms on pairs)
            hipLaunchKernelGGL(k_msm_bucket_bits1q, dim3(G1), dim3(256), 0, st, d_offs, p.B, p.nl, ws.head.as<int32_t>(), (size_t)p.G,
                               ws.cont.as<int32_t>(), (size_t)p.nl, G1, ws.chunkS.as<int32_t>(), (size_t)7 * G1, d_out);
        else
            hipLaunchKernelGGL(k_msm_bucket_bits1p, dim3(G1), dim3(128), 0, st, d_offs, p.B, p.nl, ws.head.as<int32_t>(), (size_t)p.G,
                               ws.cont.as<int32_t>(), (size_t)p.nl, G1, ws.chunkS.as<int32_t>(), (size_t)7 * G1, d_out);
        KZG_MARK(6);
        if (p.polys && p.c == 7) {
            hipLaunchKernelGGL(k_batch_finish, dim3((p.polys + 63) / 64), dim3(64), 0, st, ws.chunkS.as<int32_t>(), (size_t)7 * G1, G1, d_out);
            n_out = p.polys;
        } else if (G1 == 1) {
            n_out = 7;
        } else {
            const uint32_t waves2 = 7 * G1p;
            if (p.quad)
                hipLaunchKernelGGL(k_red_bits2q, dim3(waves2), dim3(256), 0, st, ws.chunkS.as<int32_t>(), (size_t)7 * G1, G1, G1p, d_out);
            else
                hipLaunchKernelGGL(k_red_bits2p, dim3((waves2 + 1) / 2), dim3(128), 0, st, ws.chunkS.as<int32_t>(), (size_t)7 * G1, G1, G1p, d_out);
            n_out = 13 * G1p;
        }
    } else {
        const uint32_t gg = (p.G + 255) / 256;
        hipLaunchKernelGGL(k_msm_bucket_fin, dim3(gg), dim3(256), 0, st, d_offs, p.G, p.nl, p.m, n_chunks, ws.head.as<int32_t>(), (size_t)p.G,
                           ws.cont.as<int32_t>(), (size_t)p.nl, ws.bucket.as<int32_t>(), (size_t)p.G);
        const uint32_t gc = (n_chunks + 255) / 256;
        KZG_MARK(6);
        hipLaunchKernelGGL(k_red_chunk_sums, dim3(gc), dim3(256), 0, st, ws.bucket.as<int32_t>(), (size_t)p.G, n_chunks, p.m,
                           ws.chunkS.as<int32_t>(), (size_t)n_chunks);
        hipLaunchKernelGGL(k_red_suffix_scan, dim3(n_windows), dim3(p.T), 0, st, ws.chunkS.as<int32_t>(), ws.chunkTmp.as<int32_t>(),
                           (size_t)n_chunks, p.T);
        hipLaunchKernelGGL(k_red_chunk_running, dim3(gc), dim3(256), 0, st, ws.bucket.as<int32_t>(), (size_t)p.G,
                           ws.chunkS.as<int32_t>(), (size_t)n_chunks, n_chunks, p.T, p.m, ws.chunkA.as<int32_t>());
        hipLaunchKernelGGL(k_red_window_sum, dim3(n_windows), dim3(p.T), 0, st, ws.chunkA.as<int32_t>(), (size_t)n_chunks, p.T,
                           d_out);
        n_out = n_windows;
    }
    KZG_MARK(7);
    phases.end();
    KZG_HIP_TRY(ctx, hipGetLastError());
    // this launch owns [out_off, out_off + out_cap) of the pinned result buffer (one MSM_PART_OUT window per part of a multi-part MSM)
    if (n_out > out_cap || out_off + n_out > MSM_MAX_OUT) {
        (void)hipStreamSynchronize(st);
        ctx->last_error = "MSM result points exceed this launch's window of the result buffer";
        return KZG_ERR_INVALID_ARG;
    }
    if (prof)                                             // sorted entries = mixed additions of this launch (NAF mode: data dependent)
        KZG_HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(ws.pinned_out) + (size_t)MSM_MAX_OUT * 128 + (out_off / MSM_PART_OUT) * 4, d_offs + p.G, 4,
                                        hipMemcpyDeviceToHost, st));
    if (!ws.ev_done) KZG_HIP_TRY(ctx, hipEventCreateWithFlags(&ws.ev_done, hipEventDisableTiming));
    KZG_HIP_TRY(ctx, hipEventRecord(ws.ev_done, st));        // msm_finish waits for THIS launch, not for the stream: a later MSM may already be queued behind it
#undef KZG_MARK
    pend->p = p;
    pend->n_out = n_out;
    pend->batch = batch;
    pend->n = n;
    pend->out_off = out_off;
    pend->profiled = true;
    return KZG_OK;
}

static int32_t msm_finish(kzg_ctx* ctx, MsmWorkspace& ws, hipStream_t st, const Pending& pend, kzg_host::Xyzz* result) {
    const Plan& p = pend.p;
    const uint32_t n_out = pend.n_out, batch = pend.batch;
    const uint32_t G1 = p.B / 64, G1p = (G1 + 63) / 64;
    (void)st;
    {
        RoctxRange range_wait("kzg:msm:wait");
        KZG_HIP_TRY(ctx, hipEventSynchronize(ws.ev_done));
    }
    RoctxRange range_epi("kzg:msm:host epilogue");
    if (ctx->profiling && ws.ev_ready && pend.profiled) {
        for (int i = 0; i < 7; ++i) {
            float ms = 0;
            KZG_HIP_TRY(ctx, hipEventElapsedTime(&ms, ws.ev[i], ws.ev[i + 1]));
            ws.phase_ms[i] += ms;
        }
        float ms = 0;
        KZG_HIP_TRY(ctx, hipEventElapsedTime(&ms, ws.ev[0], ws.ev[7]));
        ws.phase_ms[7] += ms;
        ws.profiled_launches += 1;
        ws.profiled_pairs += pend.n * batch;
        ws.profiled_entries += reinterpret_cast<const uint32_t*>(static_cast<const char*>(ws.pinned_out) + (size_t)MSM_MAX_OUT * 128)[pend.out_off / MSM_PART_OUT];
    }

    // host epilogue on O(100) points
    using kzg_host::Xyzz;
    static thread_local std::vector<Xyzz> vals_store;
    if (vals_store.size() < n_out) vals_store.resize(n_out);
    Xyzz* vals = vals_store.data();
    const uint64_t* w = reinterpret_cast<const uint64_t*>(ws.pinned_out) + 16 * (size_t)pend.out_off;
    for (uint32_t i = 0; i < n_out; ++i) memcpy(&vals[i], w + 16 * i, 128);
    if (!p.tables) {
        // sum_w 2^(c w) S_w: <= 255 doublings + W additions per MSM on the host (~0.1 ms).  The MSMs of a batch -- the three linear combinations of batch
        // verification -- take theirs side by side on the host pool (round 6: 0.31 -> 0.11 ms of the 1.7 ms verification core)
        if (batch > 1) host_parallel_for(batch, [&](size_t b) { result[b] = kzg_host::horner_windows(vals + b * p.W, p.W, p.c); });
        else result[0] = kzg_host::horner_windows(vals, p.W, p.c);
        return KZG_OK;
    }
    if (p.bitsum) {
        Xyzz acc = vals[0];
        for (uint32_t i = 1; i < n_out; ++i) acc = kzg_host::xyzz_add(acc, vals[i]);
        result[0] = acc;
        return KZG_OK;
    }
    if (p.polys && p.c == 7) {                             // batched table mode, 64 buckets per polynomial: k_batch_finish left one commitment each
        for (uint32_t b = 0; b < p.polys; ++b) result[b] = vals[b];
        return KZG_OK;
    }
    // sum_b (b+1) V_b = T + sum_j 2^j S_j over the bits j of the 0-based bucket index, for the units [unit_lo, unit_lo + units) of 4 096
    // buckets each (one MSM: all of them; batched table mode: the units of one polynomial)
    auto epilogue = [&](uint32_t unit_lo, uint32_t units) -> Xyzz {
        Xyzz S[32];
        int nbits = 0;
        Xyzz total;
        if (G1 == 1) {
            for (int j = 0; j < 6; ++j) S[j] = vals[j];
            nbits = 6;
            total = vals[6];
        } else {
            const Xyzz* Y = vals;
            const Xyzz* X2 = vals + 6 * G1p;
            for (int j = 0; j < 6; ++j) {
                Xyzz a = kzg_host::xyzz_inf(), b = kzg_host::xyzz_inf();
                for (uint32_t g = unit_lo; g < unit_lo + units; ++g) { a = kzg_host::xyzz_add(a, Y[j * G1p + g]); b = kzg_host::xyzz_add(b, X2[j * G1p + g]); }
                S[j] = a;
                S[6 + j] = b;
            }
            nbits = 12;
            total = kzg_host::xyzz_inf();
            for (uint32_t g = unit_lo; g < unit_lo + units; ++g) total = kzg_host::xyzz_add(total, X2[6 * G1p + g]);
            for (int i = 0; (1u << i) < units; ++i) {
                Xyzz a = kzg_host::xyzz_inf();
                for (uint32_t g = 0; g < units; ++g) if ((g >> i) & 1u) a = kzg_host::xyzz_add(a, X2[6 * G1p + unit_lo + g]);
                S[nbits++] = a;
            }
        }
        if (p.fused && G1 > 1) {                          // the fused level's group g holds the buckets gp * G1 + g: its six sums are the TOP six index bits,
            Xyzz Sk[32];                                  // the second level's the low log2(G1)
            const int lg = ilog2_floor(G1);
            for (int k = 0; k < 6; ++k) Sk[lg + k] = S[k];
            for (int j = 0; j < lg; ++j) Sk[j] = S[6 + j];
            nbits = lg + 6;
            for (int j = 0; j < nbits; ++j) S[j] = Sk[j];
        }
        if (p.naf) {                                      // the bucket index is the key rotated by six bits (naf.h naf_bucket)
            Xyzz Sk[32];
            for (int t = 0; t < nbits; ++t) Sk[naf_key_bit_of_bucket_bit(t, nbits)] = S[t];
            for (int j = 0; j < nbits; ++j) S[j] = Sk[j];
        }
        Xyzz acc = kzg_host::xyzz_inf();
        for (int j = nbits - 1; j >= 0; --j) { acc = kzg_host::xyzz_dbl(acc); acc = kzg_host::xyzz_add(acc, S[j]); }
        if (p.naf) acc = kzg_host::xyzz_dbl(acc);         // bucket b holds the odd digit 2 b + 1: sum_b (2 b + 1) V_b = 2 sum_b b V_b + T
        return kzg_host::xyzz_add(acc, total);
    };
    if (p.polys) {                                         // batched table mode with whole units per polynomial
        const uint32_t units = (1u << (p.c - 1)) / 4096u;
        for (uint32_t b = 0; b < p.polys; ++b) result[b] = epilogue(b * units, units);
        return KZG_OK;
    }
    *result = epilogue(0, G1p);
    return KZG_OK;
}
// ---- two-slot asynchronous form: begin enqueues, end waits and runs the host epilogue ---------------------------------
// Pairs per launch of an MSM over `bases`.  Tables more than 2^24 / W points apart (an SRS beyond 2^20 points at c = 17) are walked
// in power-of-two chunks whose COMPACT indices fit the two-level sort (make_plan): a 2^22-point commitment is four 2^20 launches.
static size_t msm_launch_len(const MsmBases& bases) {
    if (bases.naf) return MSM_MAX_LAUNCH;
    if (bases.table_stride != 0 && (size_t)bases.W * bases.table_stride > ((size_t)1 << SORT2_IDX_BITS)) {
        size_t cpow = 1;
        while (((size_t)bases.W * cpow * 2) <= ((size_t)1 << SORT2_IDX_BITS)) cpow *= 2;
        if (cpow >= ((size_t)1 << 16)) return cpow;
    }
    return MSM_MAX_LAUNCH;
}
// all launches of one MSM of n <= MSM_MAX_LAUNCH pairs on `st` / `ws`
static int32_t msm_enqueue_parts(kzg_ctx* ctx, MsmWorkspace& ws, hipStream_t st, const MsmBases& bases, const uint4* d_scalars, size_t n,
                                 MsmPending* mp) {
    const size_t chunk = msm_launch_len(bases);
    const size_t parts = (n + chunk - 1) / chunk;
    if (parts > MSM_MAX_PARTS) return KZG_ERR_TOO_LARGE;
    mp->n_parts = 0;
    for (size_t k = 0, off = 0; off < n; off += chunk, ++k) {
        MsmBases b = bases;
        b.points = bases.points + 4 * off;
        int32_t rc = msm_enqueue(ctx, ws, st, b, d_scalars + 2 * off, std::min(chunk, n - off), 1, &mp->part[k], parts > 1 ? (uint32_t)k * MSM_PART_OUT : 0u,
                                 parts > 1 ? MSM_PART_OUT : MSM_MAX_OUT);
        if (rc != KZG_OK) { if (k) (void)hipStreamSynchronize(st); return rc; }
        if (k) mp->part[k - 1].profiled = false;
        mp->n_parts = (uint32_t)k + 1;
    }
    return KZG_OK;
}
static int32_t msm_finish_parts(kzg_ctx* ctx, MsmWorkspace& ws, hipStream_t st, const MsmPending& mp, kzg_host::Xyzz* total) {
    *total = kzg_host::xyzz_inf();
    for (uint32_t k = 0; k < mp.n_parts; ++k) {
        kzg_host::Xyzz part;
        int32_t rc = msm_finish(ctx, ws, st, mp.part[k], &part);
        if (rc != KZG_OK) return rc;
        *total = k ? kzg_host::xyzz_add(*total, part) : part;
    }
    return KZG_OK;
}

int32_t msm_slot_stream(kzg_ctx* ctx, int slot, hipStream_t* out) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS) return KZG_ERR_INVALID_ARG;
    // One stream per slot.  How much a third / fourth MSM in flight gains depends on how HIP maps the streams onto its hardware
    // queues (4 by default, shared with every other stream of the process): at 2^17 pairs per MSM, depth 2 gives 0.43-0.45 ms per
    // MSM, depth 3 0.37-0.38 (0.5 in one mapping), depth 4 anything from 0.34 to 0.46.  Sharing streams between slots (2 or 3
    // streams for 4 slots, any assignment) made depth 4 mapping-independent but no faster than depth 2-3 (tools/queue_probe.py).
    if (slot == 0) { *out = ctx->stream; return KZG_OK; }
    if (!ctx->stream_x[slot - 1]) KZG_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream_x[slot - 1], hipStreamNonBlocking));
    *out = ctx->stream_x[slot - 1];
    return KZG_OK;
}

int32_t msm_begin(kzg_ctx* ctx, int slot, const MsmBases& bases, const void* d_scalars, size_t n) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;     // slot still in flight
    if (n == 0) return KZG_ERR_INVALID_ARG;
    if (n > 4 * MSM_MAX_LAUNCH) return KZG_ERR_TOO_LARGE;             // 2^26 pairs: MSM_MAX_PARTS launches of 2^20 (table mode) or four of 2^24
    hipStream_t st = nullptr;
    { int32_t rc0 = msm_slot_stream(ctx, slot, &st); if (rc0 != KZG_OK) return rc0; }
    MsmPending* pend = new (std::nothrow) MsmPending();
    if (!pend) return KZG_ERR_DEVICE;
    int32_t rc = msm_enqueue_parts(ctx, ctx->slot_msm(slot), st, bases, reinterpret_cast<const uint4*>(d_scalars), n, pend);
    if (rc != KZG_OK) { delete pend; return rc; }
    ctx->slot_pending[slot] = pend;
    return KZG_OK;
}

int32_t msm_end(kzg_ctx* ctx, int slot, uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || !ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;    // nothing in flight
    MsmPending* pend = ctx->slot_pending[slot];
    if (pend->n_parts && pend->part[0].p.polys) return KZG_ERR_INVALID_ARG;                            // a batched launch: kzg_msm_g1_srs_end_batch collects it
    ctx->slot_pending[slot] = nullptr;
    kzg_host::Xyzz total;
    hipStream_t st = nullptr;
    (void)msm_slot_stream(ctx, slot, &st);
    int32_t rc = msm_finish_parts(ctx, ctx->slot_msm(slot), st, *pend, &total);
    delete pend;
    if (rc != KZG_OK) return rc;
    if (out_xyzz) memcpy(out_xyzz, &total, 128);
    if (out_xy) kzg_host::xyzz_to_affine(total, out_xy, out_inf);
    return KZG_OK;
}

// polynomials one batched launch takes (batched table mode): 2^(c-1) buckets each in one 2^16-bucket array, 2^24 pairs at most
size_t msm_batch_capacity(size_t n) {
    if (n == 0) return 0;
    const int cb = batch_bucket_bits(n);
    return std::min<size_t>(std::min<size_t>(MSM_BATCH_POLYS_MAX, (size_t)65536 >> (cb - 1)), MSM_MAX_LAUNCH / n);
}

// Asynchronous form of ONE batched launch: `count` polynomials of n scalars each (separate device buffers) over the same per-bit
// tables on `slot`; msm_end_batch waits and returns count results (affine and / or XYZZ partials).
int32_t msm_begin_batch(kzg_ctx* ctx, int slot, const MsmBases& bases, const void* const* d_scalars, size_t n, size_t count) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    if (!bases.naf || n == 0 || count == 0 || count > (size_t)MSM_BATCH_PTRS || count > msm_batch_capacity(n)) return KZG_ERR_INVALID_ARG;
    hipStream_t st = nullptr;
    { int32_t rc0 = msm_slot_stream(ctx, slot, &st); if (rc0 != KZG_OK) return rc0; }
    MsmPending* pend = new (std::nothrow) MsmPending();
    if (!pend) return KZG_ERR_DEVICE;
    PolyPtrs ptrs{};
    for (size_t k = 0; k < count; ++k) ptrs.p[k] = reinterpret_cast<const uint4*>(d_scalars[k]);
    int32_t rc = msm_enqueue(ctx, ctx->slot_msm(slot), st, bases, ptrs.p[0], n * count, 1, &pend->part[0], 0, MSM_MAX_OUT, (uint32_t)count, &ptrs);
    if (rc != KZG_OK) { delete pend; return rc; }
    pend->n_parts = 1;
    ctx->slot_pending[slot] = pend;
    return KZG_OK;
}
int32_t msm_end_batch(kzg_ctx* ctx, int slot, size_t count, uint64_t* out_xy, uint8_t* out_inf, uint64_t* out_xyzz) {
    if (slot < 0 || slot >= KZG_NUM_SLOTS || !ctx->slot_pending[slot]) return KZG_ERR_INVALID_ARG;
    MsmPending* pend = ctx->slot_pending[slot];
    if (pend->n_parts != 1 || pend->part[0].p.polys != count) return KZG_ERR_INVALID_ARG;       // (not a batched launch of that size: left in flight)
    ctx->slot_pending[slot] = nullptr;
    static thread_local std::vector<kzg_host::Xyzz> res;
    res.resize(count);
    hipStream_t st = nullptr;
    (void)msm_slot_stream(ctx, slot, &st);
    int32_t rc = msm_finish(ctx, ctx->slot_msm(slot), st, pend->part[0], res.data());
    delete pend;
    if (rc != KZG_OK) return rc;
    if (out_xyzz) memcpy(out_xyzz, res.data(), count * 128);
    if (out_xy) {
        kzg_host::xyzz_batch_to_affine(res.data(), count, out_xy);
        if (out_inf)
            for (size_t i = 0; i < count; ++i) {
                const uint64_t* q = out_xy + 8 * i;
                out_inf[i] = (q[0] | q[1] | q[2] | q[3] | q[4] | q[5] | q[6] | q[7]) == 0 ? 1 : 0;
            }
    }
    return KZG_OK;
}

#ifdef KZG_ACC_STAMPS
}  // namespace kzg
// diagnostic build only: per-wave {start, end (100 MHz ticks), HW_ID, XCC_ID} of the last accumulate launch of slot 0
extern "C" int32_t kzg_debug_acc_stamps(kzg_ctx* ctx, uint64_t* out, size_t n_waves, size_t nl) {
    if (hipMemcpy(out, static_cast<const char*>(ctx->msm.cont.p) + nl * 36 * 4, n_waves * 64, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return 0;
}
namespace kzg {
#endif

void msm_drop_slots(kzg_ctx* ctx) {
    for (int s = 0; s < KZG_NUM_SLOTS; ++s) { delete ctx->slot_pending[s]; ctx->slot_pending[s] = nullptr; }
}

int32_t points_wire_to_device(kzg_ctx* ctx, const uint4* d_wire, uint4* d_out, size_t n, uint32_t* d_off_curve_flag) {
    if (n == 0) return KZG_OK;
    if (d_off_curve_flag) hipLaunchKernelGGL(k_points_wire_to_device_checked, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_wire, d_out, n, d_off_curve_flag);
    else hipLaunchKernelGGL(k_points_wire_to_device, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_wire, d_out, n);
    KZG_HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

// Negative result kept for the record (measured on MI355X at 2^20, removed from the code): cutting one large table-mode MSM
// in two halves on two streams, so that the sort / bucket reduction of one half runs beside the accumulate of the other, took
// 2.36 ms against 2.09 ms unsplit -- inside ONE MSM the halves' small kernels stretch 3-4x and the accumulates slow each other.
// Overlap pays across INDEPENDENT MSMs instead (the two slots below).
int32_t msm_run(kzg_ctx* ctx, const MsmBases& bases, const void* d_scalars, size_t n,
                uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    if (ctx->slot_pending[0]) {           // the synchronous calls share slot 0's workspace and result buffer
        ctx->last_error = "a kzg_*_begin on slot 0 is still in flight: call kzg_msm_g1_srs_end(ctx, 0, ..) first";
        return KZG_ERR_INVALID_ARG;
    }
    kzg_host::Xyzz total = kzg_host::xyzz_inf();
    const uint4* sc = reinterpret_cast<const uint4*>(d_scalars);
    static thread_local MsmPending mp;
    for (size_t off = 0; off < n; off += MSM_MAX_LAUNCH) {
        const size_t len = std::min(MSM_MAX_LAUNCH, n - off);
        MsmBases b = bases;
        b.points = bases.points + 4 * off;
        kzg_host::Xyzz part;
        int32_t rc = msm_enqueue_parts(ctx, ctx->msm, ctx->stream, b, sc + 2 * off, len, &mp);
        if (rc == KZG_OK) rc = msm_finish_parts(ctx, ctx->msm, ctx->stream, mp, &part);
        if (rc != KZG_OK) return rc;
        total = off ? kzg_host::xyzz_add(total, part) : part;
    }
    if (out_xyzz) memcpy(out_xyzz, &total, 128);
    if (out_xy) kzg_host::xyzz_to_affine(total, out_xy, out_inf);
    return KZG_OK;
}

// `batch` independent MSMs of n pairs each over concatenated caller bases (generic mode): one kernel sequence.
int32_t msm_run_batch(kzg_ctx* ctx, const uint4* d_points, const void* d_scalars, size_t n, uint32_t batch,
                      uint64_t* out_xy /* batch x 8 */, uint8_t* out_inf /* batch */) {
    if (n > MSM_MAX_LAUNCH / batch) return KZG_ERR_TOO_LARGE;
    if (ctx->slot_pending[0]) {
        ctx->last_error = "a kzg_*_begin on slot 0 is still in flight: call kzg_msm_g1_srs_end(ctx, 0, ..) first";
        return KZG_ERR_INVALID_ARG;
    }
    MsmBases b;
    b.points = d_points;
    kzg_host::Xyzz res[64];
    if (batch > 64) return KZG_ERR_INVALID_ARG;
    Pending pend;
    int32_t rc = msm_enqueue(ctx, ctx->msm, ctx->stream, b, reinterpret_cast<const uint4*>(d_scalars), n, batch, &pend);
    if (rc == KZG_OK) rc = msm_finish(ctx, ctx->msm, ctx->stream, pend, res);
    if (rc != KZG_OK) return rc;
    for (uint32_t i = 0; i < batch; ++i) kzg_host::xyzz_to_affine(res[i], out_xy + 8 * i, out_inf ? out_inf + i : nullptr);
    return KZG_OK;
}

// `polys` MSMs of n pairs each over the SAME bases (the first n points of an SRS with per-bit tables): the commitments of `polys`
// polynomials in one kernel sequence per MSM_BATCH_POLYS_MAX polynomials.  d_scalars: polys x n wire scalars, polynomial after polynomial.
int32_t msm_run_batch_tables(kzg_ctx* ctx, const MsmBases& bases, const void* d_scalars, size_t n, size_t polys, uint64_t* out_xy, uint8_t* out_inf) {
    if (!bases.naf || n == 0) return KZG_ERR_INVALID_ARG;
    for (int sl = 0; sl < KZG_NUM_SLOTS; ++sl)
        if (ctx->slot_pending[sl]) {
            ctx->last_error = "a kzg_*_begin is still in flight: the batched commitments use the slots' streams and workspaces themselves";
            return KZG_ERR_INVALID_ARG;
        }
    // polynomials per launch: 2^(c-1) buckets each in one 2^16-bucket array, 2^24 pairs at most
    const int cb = batch_bucket_bits(n);
    const size_t per = std::min<size_t>(std::min<size_t>(MSM_BATCH_POLYS_MAX, (size_t)65536 >> (cb - 1)), MSM_MAX_LAUNCH / n);
    if (per == 0) return KZG_ERR_TOO_LARGE;
    // The launches run as a software pipeline over the slots (their own streams and workspaces): launch k + 1 .. k + 2 are enqueued
    // before launch k is collected, so the latency-bound kernels of one overlap the accumulate kernel of another -- a launch of 2^17
    // scalars takes 0.42 ms alone and 0.22 ms with three in flight (section 6b of DESIGN.md).
    constexpr int DEPTH = 3;
    static_assert(DEPTH <= KZG_NUM_SLOTS, "one slot per launch in flight");
    Pending pend[DEPTH];
    size_t first[DEPTH] = {}, cnt[DEPTH] = {};
    bool busy[DEPTH] = {};
    static thread_local std::vector<kzg_host::Xyzz> res;
    res.resize(std::min(per, polys));
    const uint4* sc = reinterpret_cast<const uint4*>(d_scalars);
    auto collect = [&](int s) -> int32_t {
        hipStream_t st = nullptr;
        (void)msm_slot_stream(ctx, s, &st);
        int32_t rc = msm_finish(ctx, ctx->slot_msm(s), st, pend[s], res.data());
        busy[s] = false;
        if (rc != KZG_OK) return rc;
        kzg_host::xyzz_batch_to_affine(res.data(), cnt[s], out_xy + 8 * first[s]);
        if (out_inf)
            for (size_t i = 0; i < cnt[s]; ++i) {
                const uint64_t* q = out_xy + 8 * (first[s] + i);
                out_inf[first[s] + i] = (q[0] | q[1] | q[2] | q[3] | q[4] | q[5] | q[6] | q[7]) == 0 ? 1 : 0;
            }
        return KZG_OK;
    };
    int32_t rc = KZG_OK;
    size_t launch = 0;
    for (size_t done = 0; done < polys && rc == KZG_OK; done += per, ++launch) {
        const int s = (int)(launch % DEPTH);
        if (busy[s]) rc = collect(s);
        if (rc != KZG_OK) break;
        const size_t k = std::min(per, polys - done);
        hipStream_t st = nullptr;
        rc = msm_slot_stream(ctx, s, &st);
        if (rc != KZG_OK) break;
        rc = msm_enqueue(ctx, ctx->slot_msm(s), st, bases, sc + 2 * done * n, n * k, 1, &pend[s], 0, MSM_MAX_OUT, (uint32_t)k);
        if (rc != KZG_OK) break;
        first[s] = done; cnt[s] = k; busy[s] = true;
    }
    for (size_t q = 0; q < DEPTH; ++q) {                   // drain in launch order (also after an error: nothing stays in flight)
        const int s = (int)((launch + q) % DEPTH);
        if (!busy[s]) continue;
        const int32_t r2 = collect(s);
        if (rc == KZG_OK) rc = r2;
    }
    return rc;
}

}  // namespace kzg
