// msm.hip — host orchestration of the G1 MSM kernels (msm_kernels.h).
// Replaces `G1Projective::msm(..)` + `.into_affine()` at prover/src/kzg.rs:100-101, :121-122 and
// primitives/src/helpers.rs:332-336.
#include "engine.h"
#include "msm_kernels.h"
#include "host_curve.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kzg {

void MsmWorkspace::release() {
    DeviceBuffer* all[] = {&scalars, &bases, &bases_wire, &digits, &sorted, &count, &cursor, &offs, &block_sums,
                           &seg_bucket, &segsum, &bucket, &chunkS, &chunkTmp, &chunkA, &out_wire};
    for (auto* b : all) b->release();
    if (pinned_out) { (void)hipHostFree(pinned_out); pinned_out = nullptr; }
    if (ev_ready) { for (auto& e : ev) (void)hipEventDestroy(e); ev_ready = false; }
}

static int ilog2_floor(size_t n) { int k = 0; while ((n >> (k + 1)) != 0) ++k; return k; }

static MsmPlan make_plan(const kzg_ctx* ctx, size_t n) {
    MsmPlan p;
    p.n = (uint32_t)n;
    int c = ctx->msm_c_override;
    if (c == 0) {
        const char* env = getenv("KZG_MSM_C");
        if (env) c = atoi(env);
    }
    if (c == 0) c = std::min(14, std::max(4, ilog2_floor(n) - 6));
    c = std::min(16, std::max(2, c));
    p.c = c;
    p.W = (255 + c - 1) / c;
    p.B = 1u << (c - 1);
    p.G = (uint32_t)p.W * p.B;
    int L = ctx->msm_seg_override;
    if (L == 0) {
        const char* env = getenv("KZG_MSM_SEG");
        if (env) L = atoi(env);
    }
    if (L <= 0) L = 64;
    p.L = (uint32_t)L;
    p.T = std::min<uint32_t>(p.B, RED_T);
    p.m = p.B / p.T;
    size_t entries = (size_t)p.W * n;
    p.segcap = (uint32_t)(entries / p.L + std::min<size_t>(p.G, entries) + 1);
    return p;
}

// Largest number of pairs one launch takes: W * n must fit the 32-bit positions of the sort.
static const size_t MSM_MAX_LAUNCH = (size_t)1 << 24;

static int32_t msm_launch(kzg_ctx* ctx, const uint4* d_points, const uint4* d_scalars, size_t n, kzg_host::Xyzz* result) {
    MsmPlan p = make_plan(ctx, n);
    MsmWorkspace& ws = ctx->msm;
    hipStream_t st = ctx->stream;
    const size_t entries = (size_t)p.W * n;
    const uint32_t n_chunks = (uint32_t)p.W * p.T;
    const uint32_t nb = (p.G + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > SCAN_TILE) return KZG_ERR_INVALID_ARG;

    KZG_HIP_TRY(ctx, ws.digits.reserve(entries * 4));
    KZG_HIP_TRY(ctx, ws.sorted.reserve(entries * 4));
    KZG_HIP_TRY(ctx, ws.count.reserve((size_t)p.G * 4));
    KZG_HIP_TRY(ctx, ws.cursor.reserve((size_t)p.G * 4));
    KZG_HIP_TRY(ctx, ws.offs.reserve(((size_t)p.G + 1) * 8));
    KZG_HIP_TRY(ctx, ws.block_sums.reserve((size_t)SCAN_TILE * 8));
    KZG_HIP_TRY(ctx, ws.seg_bucket.reserve((size_t)p.segcap * 4));
    KZG_HIP_TRY(ctx, ws.segsum.reserve((size_t)p.segcap * 36 * 4));
    KZG_HIP_TRY(ctx, ws.bucket.reserve((size_t)p.G * 36 * 4));
    KZG_HIP_TRY(ctx, ws.chunkS.reserve((size_t)n_chunks * 36 * 4));
    KZG_HIP_TRY(ctx, ws.chunkTmp.reserve((size_t)n_chunks * 36 * 4));
    KZG_HIP_TRY(ctx, ws.chunkA.reserve((size_t)n_chunks * 36 * 4));
    KZG_HIP_TRY(ctx, ws.out_wire.reserve((size_t)p.W * 32 * 4));
    if (!ws.pinned_out) KZG_HIP_TRY(ctx, hipHostMalloc(&ws.pinned_out, 64 * 32 * 4 * 4, hipHostMallocDefault));

    const bool prof = ctx->profiling;
    if (prof && !ws.ev_ready) {
        for (auto& e : ws.ev) KZG_HIP_TRY(ctx, hipEventCreate(&e));
        ws.ev_ready = true;
    }
#define KZG_MARK(i) do { if (prof) KZG_HIP_TRY(ctx, hipEventRecord(ws.ev[i], st)); } while (0)

    KZG_MARK(0);
    KZG_HIP_TRY(ctx, hipMemsetAsync(ws.count.p, 0, (size_t)p.G * 4, st));
    KZG_HIP_TRY(ctx, hipMemsetAsync(ws.cursor.p, 0, (size_t)p.G * 4, st));

    const uint32_t gn = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_msm_digits, dim3(gn), dim3(256), 0, st, d_scalars, p.n, p.c, p.W, p.B,
                       ws.digits.as<uint32_t>(), ws.count.as<uint32_t>());
    KZG_MARK(1);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, p.L,
                       ws.block_sums.as<unsigned long long>());
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, st, ws.block_sums.as<unsigned long long>(), nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_THREADS), 0, st, ws.count.as<uint32_t>(), p.G, p.L,
                       ws.block_sums.as<unsigned long long>(), ws.offs.as<unsigned long long>());
    KZG_MARK(2);
    hipLaunchKernelGGL(k_msm_scatter, dim3(gn), dim3(256), 0, st, ws.digits.as<uint32_t>(), p.n, p.W, p.B,
                       ws.offs.as<unsigned long long>(), ws.cursor.as<uint32_t>(), ws.sorted.as<uint32_t>());
    const uint32_t gg = (p.G + 255) / 256;
    KZG_MARK(3);
    hipLaunchKernelGGL(k_msm_segments, dim3(gg), dim3(256), 0, st, ws.offs.as<unsigned long long>(), p.G,
                       ws.seg_bucket.as<uint32_t>());
    const uint32_t gs = (p.segcap + 255) / 256;
    KZG_MARK(4);
    hipLaunchKernelGGL(k_msm_accumulate, dim3(gs), dim3(256), 0, st, d_points, ws.sorted.as<uint32_t>(),
                       ws.seg_bucket.as<uint32_t>(), ws.offs.as<unsigned long long>(), p.G, p.L,
                       ws.segsum.as<int32_t>(), (size_t)p.segcap);
    KZG_MARK(5);
    hipLaunchKernelGGL(k_msm_bucket_fin, dim3(gg), dim3(256), 0, st, ws.offs.as<unsigned long long>(), p.G, p.m, n_chunks,
                       ws.segsum.as<int32_t>(), (size_t)p.segcap, ws.bucket.as<int32_t>(), (size_t)p.G);
    const uint32_t gc = (n_chunks + 255) / 256;
    KZG_MARK(6);
    hipLaunchKernelGGL(k_red_chunk_sums, dim3(gc), dim3(256), 0, st, ws.bucket.as<int32_t>(), (size_t)p.G, n_chunks, p.m,
                       ws.chunkS.as<int32_t>(), (size_t)n_chunks);
    hipLaunchKernelGGL(k_red_suffix_scan, dim3(p.W), dim3(p.T), 0, st, ws.chunkS.as<int32_t>(), ws.chunkTmp.as<int32_t>(),
                       (size_t)n_chunks, p.T);
    hipLaunchKernelGGL(k_red_chunk_running, dim3(gc), dim3(256), 0, st, ws.bucket.as<int32_t>(), (size_t)p.G,
                       ws.chunkS.as<int32_t>(), (size_t)n_chunks, n_chunks, p.T, p.m, ws.chunkA.as<int32_t>());
    hipLaunchKernelGGL(k_red_window_sum, dim3(p.W), dim3(p.T), 0, st, ws.chunkA.as<int32_t>(), (size_t)n_chunks, p.T,
                       ws.out_wire.as<uint32_t>());
    KZG_MARK(7);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ws.pinned_out, ws.out_wire.p, (size_t)p.W * 128, hipMemcpyDeviceToHost, st));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(st));
    if (prof) {
        for (int i = 0; i < 7; ++i) {
            float ms = 0;
            KZG_HIP_TRY(ctx, hipEventElapsedTime(&ms, ws.ev[i], ws.ev[i + 1]));
            ws.phase_ms[i] += ms;
        }
        float ms = 0;
        KZG_HIP_TRY(ctx, hipEventElapsedTime(&ms, ws.ev[0], ws.ev[7]));
        ws.phase_ms[7] += ms;
        ws.profiled_launches += 1;
        ws.profiled_pairs += n;
    }
#undef KZG_MARK

    // Horner over the window sums: at most 255 doublings on the host, beside the D2H copy
    kzg_host::Xyzz sums[128];
    const uint64_t* w = reinterpret_cast<const uint64_t*>(ws.pinned_out);
    for (int i = 0; i < p.W; ++i) memcpy(&sums[i], w + 16 * i, 128);
    *result = kzg_host::horner_windows(sums, p.W, p.c);
    return KZG_OK;
}

int32_t points_wire_to_device(kzg_ctx* ctx, const uint4* d_wire, uint4* d_out, size_t n) {
    if (n == 0) return KZG_OK;
    hipLaunchKernelGGL(k_points_wire_to_device, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_wire, d_out, n);
    KZG_HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

int32_t msm_run(kzg_ctx* ctx, const uint4* d_points, const void* d_scalars, size_t n,
                uint64_t out_xy[8], uint8_t* out_inf, uint64_t* out_xyzz) {
    kzg_host::Xyzz total = kzg_host::xyzz_inf();
    for (size_t off = 0; off < n; off += MSM_MAX_LAUNCH) {
        size_t len = std::min(MSM_MAX_LAUNCH, n - off);
        kzg_host::Xyzz part;
        int32_t rc = msm_launch(ctx, d_points + 4 * off, reinterpret_cast<const uint4*>(d_scalars) + 2 * off, len, &part);
        if (rc != KZG_OK) return rc;
        total = kzg_host::xyzz_add(total, part);
    }
    if (out_xyzz) memcpy(out_xyzz, &total, 128);
    if (out_xy) kzg_host::xyzz_to_affine(total, out_xy, out_inf);
    return KZG_OK;
}

}  // namespace kzg
