// Micro-benchmark: is a BATCHED-AFFINE bucket accumulation faster than the XYZZ mixed addition of k_msm_accumulate?
//
// Affine + affine with Montgomery's trick costs 5M + 1S per addition (prefix product, two products on the way back, lambda, lambda^2,
// y3) = 936 v_mad_i64_i32 against the 1 467 of xyzz_madd (8M + 2S) -- IF the shared inversion and its bookkeeping are free.  This
// program measures what they cost on MI355X, on the access pattern of the real kernel (a random 64-byte gather from a table of
// `table_log` points through a per-lane index stream, as tools/ubench/acc_variants.hip V4):
//
//   madd        the loop of k_msm_accumulate: one xyzz_madd per index                                       (the baseline)
//   ba<K,INV>   level 0 of a pairwise tree: a lane owns K pairs (2K indices) per round; forward pass gathers both points of a pair,
//               multiplies the denominators x2 - x1 into a running product whose K prefixes are parked (LDS for K <= 8, a global
//               scratch plane beyond: 36 B written + 36 B read per addition); ONE inversion; backward pass gathers the points again
//               (they do not fit anywhere on chip: K x 128 B per lane), finishes the K additions and stores K packed affine sums (64 B)
//     INV = 0   one Fermat inversion per lane per round (381 products)
//     INV = 1   NO inversion (the running product stands in for its inverse: wrong values, same instruction stream): the FLOOR of the
//               scheme, whatever inversion algorithm is used and however widely it is shared
//     INV = 2   one inversion per WAVE: 6-level butterfly of lane products over ds_bpermute (6 + 6 products per lane), Fermat on the total
//   ba_seq      levels >= 1 of the tree: the two inputs are consecutive 64-byte points of the previous level (streamed, not gathered)
//
// Every variant processes the same number of ADDITIONS per lane; the figure of merit is ns per addition per SIMD (and the whole-chip
// rate beside the 14.7 G additions/s of k_msm_accumulate at 0.98 ms).  `check` compares the batched sums with xyzz_madd on the same pairs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 batch_affine.hip -o batch_affine        Run: ./batch_affine [table_log=24] [adds_per_lane=48]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rust-kzg-bn254_amd/csrc/curve.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F>
__device__ __forceinline__ void fe_inverse(Fe<F>& out, const Fe<F>& a) {     // a^(m-2), as csrc/srs.hip
    Fe<F> acc, base = a;
    fe_set_one(acc);
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = F::P32[j];
    e[0] -= 2u;
    for (int w = 0; w < 8; ++w) {
        uint32_t bits = e[w];
        for (int b = 0; b < 32; ++b) {
            if (w == 7 && b >= 30) break;
            if (bits & 1u) fe_mul(acc, acc, base);
            fe_sqr(base, base);
            bits >>= 1;
        }
    }
    out = acc;
}

__device__ __forceinline__ void load_point(uint4 (&q)[4], const uint4* __restrict__ src) { q[0] = src[0]; q[1] = src[1]; q[2] = src[2]; q[3] = src[3]; }
__device__ __forceinline__ void unpack_x(Fq& x, const uint4 (&q)[4]) { uint32_t w[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w}; fe_unpack(x, w); }
__device__ __forceinline__ void unpack_y(Fq& y, const uint4 (&q)[4]) { uint32_t w[8] = {q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w}; fe_unpack(y, w); }

__global__ void k_fill(uint4* pts, size_t n) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        uint32_t w[16];
        for (int j = 0; j < 16; ++j) w[j] = ((uint32_t)(p * 2654435761ull) ^ (uint32_t)(j * 40503u + p) ^ (uint32_t)(p >> 13) * 0x9E3779B9u) & 0x0FFFFFFFu;
        w[0] = (uint32_t)p & 0x0FFFFFFFu;                     // x differs between any two indices
        for (int j = 0; j < 4; ++j) pts[4 * p + j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
    }
}

// ---- baseline: the accumulate loop (acc_variants.hip V4) ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_madd(uint32_t* out, const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, uint32_t idx_mask, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t begin = t * (uint32_t)iters, end = begin + (uint32_t)iters, last = end - 1;
    uint32_t v = sorted[begin];
    uint32_t v1 = sorted[begin + 1 < end ? begin + 1 : last];
    uint4 q[4];
    load_point(q, points + 4 * (size_t)(v & idx_mask));
    Affine p0; unpack_x(p0.x, q); unpack_y(p0.y, q);
    Xyzz acc; xyzz_from_affine(acc, p0, 0);
    for (uint32_t e = begin; e < end; ++e) {
        const uint32_t neg = v >> 31;
        Affine p; unpack_x(p.x, q); unpack_y(p.y, q);
        v = v1;
        load_point(q, points + 4 * (size_t)(v & idx_mask));
        v1 = sorted[e + 2 < end ? e + 2 : last];
        xyzz_madd<true>(acc, p, neg);
    }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)acc.x.l[j] ^ (uint32_t)acc.y.l[j] ^ (uint32_t)acc.zz.l[j];
    out[t] = x;
}

// ---- batched affine --------------------------------------------------------------------------------------------------------------------
// prefix product i of lane tl: LDS plane (K <= 8) or global plane
template <int K, bool GLOBAL>
struct Prefix {
    int32_t* lds; int32_t* glob; size_t lanes; uint32_t t, tl;
    __device__ __forceinline__ void put(int i, const Fq& v) {
#pragma unroll
        for (int j = 0; j < NL; ++j) { if (GLOBAL) glob[((size_t)i * NL + j) * lanes + t] = v.l[j]; else lds[(i * NL + j) * 256 + tl] = v.l[j]; }
    }
    __device__ __forceinline__ void get(int i, Fq& v) {
#pragma unroll
        for (int j = 0; j < NL; ++j) v.l[j] = GLOBAL ? glob[((size_t)i * NL + j) * lanes + t] : lds[(i * NL + j) * 256 + tl];
    }
};

template <int K, int INV, bool GLOBAL, bool SEQ, bool CHECK_MODE>
__global__ void __launch_bounds__(256)
k_ba(uint4* __restrict__ out, const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, uint32_t idx_mask, int rounds,
     int32_t* __restrict__ scratch, uint32_t* __restrict__ mismatches) {
    extern __shared__ int32_t lds_pre[];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, tl = threadIdx.x;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    Prefix<K, GLOBAL> pre{lds_pre, scratch, lanes, t, tl};
    for (int r = 0; r < rounds; ++r) {
        const uint32_t base = (t * (uint32_t)rounds + (uint32_t)r) * 2u * K;          // this round's 2K indices
        // SEQ: inputs are consecutive points of a previous level (lane-major so that a wave's loads stay within a few lines)
        auto src_of = [&](int i, int which) -> const uint4* {
            if (SEQ) return points + 4 * ((((size_t)r * 2 * K + 2 * i + which) * lanes + t) & idx_mask);     // consecutive lanes, consecutive points
            return points + 4 * (size_t)(sorted[base + 2 * i + which] & idx_mask);
        };
        auto sign_of = [&](int i, int which) -> uint32_t { return SEQ ? 0u : sorted[base + 2 * i + which] >> 31; };
        Fq run;
        fe_set_one(run);
        // forward: denominators and their running products
        uint4 qa[4], qb[4];
        load_point(qa, src_of(0, 0)); load_point(qb, src_of(0, 1));
        for (int i = 0; i < K; ++i) {
            Fq x1, x2, dx;
            unpack_x(x1, qa); unpack_x(x2, qb);
            const int nx = i + 1 < K ? i + 1 : K - 1;
            load_point(qa, src_of(nx, 0)); load_point(qb, src_of(nx, 1));               // prefetch the next pair (x only is needed, the line comes whole)
            fe_sub(dx, x2, x1); fe_norm(dx);
            fe_mul(run, run, dx);
            pre.put(i, run);
        }
        // the one inversion
        Fq inv;
        if (INV == 0) fe_inverse(inv, run);
        else if (INV == 1) inv = run;
        else {
            Fq cur = run, saved[6];
#pragma unroll
            for (int l = 0; l < 6; ++l) {
#pragma unroll
                for (int j = 0; j < NL; ++j) saved[l].l[j] = __shfl_xor(cur.l[j], 1 << l, 64);
                fe_mul(cur, cur, saved[l]);
            }
            fe_inverse(inv, cur);                                                    // the same total on every lane
#pragma unroll
            for (int l = 0; l < 6; ++l) fe_mul(inv, inv, saved[l]);                  // x the product of the other 63 lanes = 1 / this lane's product
        }
        // backward: finish the K additions
        load_point(qa, src_of(K - 1, 0)); load_point(qb, src_of(K - 1, 1));
        for (int i = K - 1; i >= 0; --i) {
            Fq x1, y1, x2, y2, dx, before, iq, dy, lam, x3, y3, tt;
            unpack_x(x1, qa); unpack_y(y1, qa); unpack_x(x2, qb); unpack_y(y2, qb);
            const uint32_t s1 = sign_of(i, 0), s2 = sign_of(i, 1);
            const int nx = i > 0 ? i - 1 : 0;
            load_point(qa, src_of(nx, 0)); load_point(qb, src_of(nx, 1));
            fe_cneg(y1, y1, s1); fe_cneg(y2, y2, s2);
            if (i == 0) fe_set_one(before); else pre.get(i - 1, before);
            fe_sub(dx, x2, x1); fe_norm(dx);
            fe_mul(iq, inv, before);                       // 1 / dx_i
            fe_mul(inv, inv, dx);                          // 1 / (dx_0 .. dx_(i-1))
            fe_sub(dy, y2, y1); fe_norm(dy);
            fe_mul(lam, dy, iq);
            fe_sqr(x3, lam);
            fe_sub(x3, x3, x1); fe_sub(x3, x3, x2); fe_norm(x3);          // (-3m, 2m)
            fe_sub(tt, x1, x3); fe_norm(tt);                              // (-2m, 4m)
            fe_mul(y3, lam, tt);
            fe_sub(y3, y3, y1); fe_norm(y3);                              // (-3m, 3m)
            if (CHECK_MODE && INV != 1) {
                // the same sum through the XYZZ formulas: x3 ZZ == X, y3 ZZZ == Y (mod m)
                Affine pa, pb; pa.x = x1; pa.y = y1; pb.x = x2; pb.y = y2;
                Xyzz acc; xyzz_from_affine(acc, pa, 0);
                xyzz_madd<true>(acc, pb, 0);
                Fq lx, ly, d1, d2;
                Fq x3n = x3, y3n = y3;
                fe_reduce(x3n); fe_reduce(y3n);
                fe_mul(lx, x3n, acc.zz); fe_mul(ly, y3n, acc.zzz);
                fe_sub(d1, lx, acc.x); fe_sub(d2, ly, acc.y);
                fe_reduce(d1); fe_reduce(d2);
                if (!fe_is_zero_mod(d1) || !fe_is_zero_mod(d2)) atomicAdd(mismatches, 1u);
            }
            // canonical packed affine result (what the next level of the tree reads)
            fe_reduce_small(x3); fe_canon(x3);
            fe_reduce_small(y3); fe_canon(y3);
            uint32_t o[16];
            fe_pack(o, x3); fe_pack(o + 8, y3);
            uint4* dst = out + 4 * ((size_t)t * rounds * K + (size_t)r * K + i);
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]); dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
            dst[2] = make_uint4(o[8], o[9], o[10], o[11]); dst[3] = make_uint4(o[12], o[13], o[14], o[15]);
        }
    }
}

struct Result { float ms; int regs; int waves_per_simd; };

template <class Kern, class... Args>
static int time_kernel(Result* res, Kern kern, int blocks, size_t lds, Args... args) {
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    if (lds > 48 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...); CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(t0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...); CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
        float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms;
    }
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, lds));
    res->ms = best; res->regs = fa.numRegs; res->waves_per_simd = occ;      // a 256-thread block = one wave per SIMD
    return 0;
}

int main(int argc, char** argv) {
    const int table_log = argc > 1 ? atoi(argv[1]) : 24;                        // 2^24 points = 1 GiB (the window tables); 28 = 16 GiB (the per-bit tables)
    const int adds = argc > 2 ? atoi(argv[2]) : 48;                             // additions per lane (k_msm_accumulate at 2^20: ~73 entries per lane)
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t table_pts = (size_t)1 << table_log;
    uint4* d_pts; CHECK(hipMalloc(&d_pts, table_pts * 64));
    // distinct, canonical-looking points (every word < 2^28, so every coordinate < 2^252 < m): the arithmetic cost does not depend on the values
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_pts, table_pts);
    CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
    const int max_blocks = cus * 3;                                             // 3 waves per SIMD of 256-thread blocks: the accumulate kernel's grid
    const size_t max_lanes = (size_t)max_blocks * 256;
    const size_t n_idx = max_lanes * (size_t)adds * 2;
    std::vector<uint32_t> s(n_idx);
    uint64_t x = 88172645463325252ull;
    for (auto& v : s) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 20) ^ (uint32_t)(x << 31); }
    uint32_t* d_sorted; CHECK(hipMalloc(&d_sorted, n_idx * 4));
    CHECK(hipMemcpy(d_sorted, s.data(), n_idx * 4, hipMemcpyHostToDevice));
    uint4* d_out; CHECK(hipMalloc(&d_out, max_lanes * (size_t)adds * 64));
    int32_t* d_scratch; CHECK(hipMalloc(&d_scratch, max_lanes * 32 * NL * 4));
    uint32_t* d_mis; CHECK(hipMalloc(&d_mis, 4)); CHECK(hipMemset(d_mis, 0, 4));
    const uint32_t mask = (uint32_t)(table_pts - 1);
    const double simds = cus * 4.0;
    printf("device: %s, %d CUs; table 2^%d points (%.1f GiB); %d additions per lane\n", prop.name, cus, table_log, table_pts * 64.0 / (1u << 30), adds);
    printf("%-44s %8s %6s %10s %14s %12s\n", "variant", "ms", "VGPRs", "waves/SIMD", "ns/add/SIMD", "G adds/s");

    auto report = [&](const char* name, const Result& r, int blocks, double adds_per_lane) {
        const double total = (double)blocks * 256.0 * adds_per_lane;
        printf("%-44s %8.3f %6d %10d %14.1f %12.2f\n", name, r.ms, r.regs, r.waves_per_simd, r.ms * 1e6 / (total / 64.0 / simds), total / (r.ms * 1e-3) / 1e9);
    };
    Result r;
    // baseline: `adds` mixed additions per lane, 3 waves per SIMD
    if (time_kernel(&r, k_madd, max_blocks, 0, (uint32_t*)d_out, (const uint4*)d_pts, (const uint32_t*)d_sorted, mask, adds)) return 1;
    report("madd (k_msm_accumulate's loop), 3 waves", r, max_blocks, adds);
    const double madd_rate = (double)max_blocks * 256.0 * adds / (r.ms * 1e-3);

#define RUN_BA(K, INV, GLOBAL, SEQ, NAME)                                                                                                   \
    do {                                                                                                                                        \
        const size_t lds = GLOBAL ? 0 : (size_t)K * NL * 256 * 4;                                                                               \
        int occ = 0;                                                                                                                            \
        auto kern = k_ba<K, INV, GLOBAL, SEQ, false>;                                                                                           \
        if (lds > 48 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, lds));                               \
        if (occ > 3) occ = 3;                                                                                                                   \
        const int blocks = cus * (occ > 0 ? occ : 1);                                                                                           \
        int rounds = (int)((size_t)adds * max_blocks / blocks / K);            /* the same total number of additions as the baseline */          \
        if (rounds < 1) rounds = 1;                                                                                                             \
        if (time_kernel(&r, kern, blocks, lds, d_out, (const uint4*)d_pts, (const uint32_t*)d_sorted, mask, rounds, d_scratch, d_mis)) return 1;  \
        r.waves_per_simd = occ;                                                                                                                 \
        report(NAME, r, blocks, (double)rounds * K);                                                                                            \
        printf("%-44s   -> %.2f x the baseline's addition rate\n", "", ((double)blocks * 256.0 * rounds * K / (r.ms * 1e-3)) / madd_rate);       \
    } while (0)

    RUN_BA(8, 1, false, false, "ba K=8  LDS     NO inversion (floor)");
    RUN_BA(16, 1, true, false, "ba K=16 global  NO inversion (floor)");
    RUN_BA(32, 1, true, false, "ba K=32 global  NO inversion (floor)");
    RUN_BA(8, 1, false, true, "ba K=8  LDS     NO inversion, streamed inputs");
    RUN_BA(32, 1, true, true, "ba K=32 global  NO inversion, streamed inputs");
    RUN_BA(8, 0, false, false, "ba K=8  LDS     Fermat per lane");
    RUN_BA(32, 0, true, false, "ba K=32 global  Fermat per lane");
    RUN_BA(8, 2, false, false, "ba K=8  LDS     Fermat per wave (butterfly)");
    RUN_BA(32, 2, true, false, "ba K=32 global  Fermat per wave (butterfly)");

    // correctness of the batched formulas against xyzz_madd on the same pairs (small run)
    {
        CHECK(hipMemset(d_mis, 0, 4));
        auto k0 = k_ba<8, 0, false, false, true>;
        auto k2 = k_ba<8, 2, false, false, true>;
        const size_t lds = (size_t)8 * NL * 256 * 4;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k0, dim3(8), dim3(256), lds, 0, d_out, (const uint4*)d_pts, (const uint32_t*)d_sorted, mask, 2, d_scratch, d_mis);
        hipLaunchKernelGGL(k2, dim3(8), dim3(256), lds, 0, d_out, (const uint4*)d_pts, (const uint32_t*)d_sorted, mask, 2, d_scratch, d_mis);
        CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
        uint32_t mis = 0; CHECK(hipMemcpy(&mis, d_mis, 4, hipMemcpyDeviceToHost));
        printf("check: %u of %d batched sums differ from xyzz_madd (per-lane and per-wave inversion)\n", mis, 2 * 8 * 256 * 2 * 8);
        if (mis) return 2;
    }
    return 0;
}
