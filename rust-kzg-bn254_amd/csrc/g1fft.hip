// g1fft.hip — `KZG::g1_ifft` (prover/src/kzg.rs:263-285): the inverse FFT whose DATA are G1 points and whose
// twiddles are Fr scalars,  L_i = n^-1 * sum_j w^(-ij) P_j  (natural order) — the Lagrange-basis SRS.
// Reference: `GeneralEvaluationDomain::<Fr>::new(n).ifft(&[G1Projective])` + n `into_affine()` inversions.
// The commit / proof path of this library never needs it (commit_eval_form == MSM(srs, IFFT(evals)));
// it is provided because it is public API with its own bench (prover/benches/bench_g1_ifft.rs) and golden vector
// (prover/tests/test-files/lagrangeG1SRS.txt).
//
// Radix-2 decimation in time on XYZZ points held as 36 limb planes: bit-reversed load, log2 n stages of n/2
// butterflies (A, B) -> (A + [w]B, A - [w]B) with [w]B a 254-bit double-and-add (skipped for w = 1), then the scaling
// by n^-1 and one inversion per point.  Work = (n/2) log2 n scalar multiplications: integer-VALU bound.
#include "engine.h"
#include "curve.h"

namespace kzg {

template <class F>
__device__ __forceinline__ void fe_inverse_fermat(Fe<F>& out, const Fe<F>& a) {
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

// internal-form Fr (|v| < 169 m) -> canonical integer words
__device__ __forceinline__ void fr_to_canonical_words(uint32_t k[8], const Fr& v) {
    Fr one_plain, c;
    fe_set_zero(one_plain);
    one_plain.l[0] = 1;
    fe_mul(c, v, one_plain);
    fe_canon(c);
    fe_pack(k, c);
}

// r = [k] * p, k canonical 256-bit words (MSB-first double-and-add)
__device__ __forceinline__ void xyzz_scalar_mul(Xyzz& r, const Xyzz& p, const uint32_t k[8]) {
    Xyzz acc;
    xyzz_set_inf(acc);
    for (int w = 7; w >= 0; --w) {
        const uint32_t bits = k[w];
        for (int b = 31; b >= 0; --b) {
            Xyzz t;
            xyzz_dbl_impl(t, acc);
            acc = t;
            if ((bits >> b) & 1u) {
                xyzz_add<true>(t, acc, p);
                acc = t;
            }
        }
    }
    r = acc;
}

__device__ __forceinline__ void tw_load(Fr& w, const NttTables& tb, uint32_t E) {
#pragma unroll
    for (int j = 0; j < NL; ++j) w.l[j] = tb.lo[(size_t)j * tb.lo_len + (E & (tb.lo_len - 1))];
    uint32_t eh = E >> tb.lo_bits;
    if (eh != 0) {
        Fr h;
#pragma unroll
        for (int j = 0; j < NL; ++j) h.l[j] = tb.hi[(size_t)j * tb.hi_len + eh];
        fe_mul(w, w, h);
    }
}

// planes[i] = P[bitrev(i)] as XYZZ
__global__ void __launch_bounds__(256)
k_g1fft_load(const uint4* __restrict__ points, uint32_t n, int log_n, int32_t* __restrict__ planes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = log_n ? (__brev(i) >> (32 - log_n)) : 0;
    Affine p;
    Xyzz v;
    if (affine_load(p, points + 4 * (size_t)j)) xyzz_from_affine(v, p, 0);
    else xyzz_set_inf(v);
    xyzz_store(planes, n, i, v);
}

// stage s (block length m = 2^s): one butterfly per thread, in place
__global__ void __launch_bounds__(256)
k_g1fft_stage(int32_t* __restrict__ planes, uint32_t n, int log_n, int s, NttTables tb_inv) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n / 2) return;
    const uint32_t half = 1u << (s - 1);
    const uint32_t j = b & (half - 1);
    const uint32_t i0 = ((b >> (s - 1)) << s) | j, i1 = i0 + half;
    Xyzz A, B, t;
    xyzz_load(A, planes, n, i0);
    xyzz_load(B, planes, n, i1);
    const uint32_t E = j << (log_n - s);                 // w_n^(-j n/m)
    if (E == 0) {
        t = B;
    } else {
        Fr w;
        tw_load(w, tb_inv, E);
        uint32_t k[8];
        fr_to_canonical_words(k, w);
        xyzz_scalar_mul(t, B, k);
    }
    Xyzz r0, r1, tn = t;
    if (!tn.inf) { fe_neg(tn.y, t.y); fe_norm(tn.y); }
    xyzz_add<true>(r0, A, t);
    xyzz_add<true>(r1, A, tn);
    xyzz_store(planes, n, i0, r0);
    xyzz_store(planes, n, i1, r1);
}

// L_i = n^-1 * planes[i], converted to affine; out in wire format (16 u32 per point)
__global__ void __launch_bounds__(256)
k_g1fft_finish(const int32_t* __restrict__ planes, uint32_t n, int log_n, uint4* __restrict__ out_wire) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz v, r;
    xyzz_load(v, planes, n, i);
    if (log_n > 0) {
        Fr ninv;
#pragma unroll
        for (int j = 0; j < NL; ++j) ninv.l[j] = (int32_t)FrParams::NINV[log_n * NL + j];
        uint32_t k[8];
        fr_to_canonical_words(k, ninv);
        xyzz_scalar_mul(r, v, k);
    } else {
        r = v;
    }
    uint32_t o[16];
    if (r.inf) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0;
    } else {
        Fq zi, t, x, y;
        fe_mul(t, r.zz, r.zzz);
        fe_inverse_fermat(zi, t);
        fe_mul(t, zi, r.zzz);
        fe_mul(x, r.x, t);
        fe_mul(t, zi, r.zz);
        fe_mul(y, r.y, t);
        fe_norm(x);
        fe_norm(y);
        fe_to_wire(o, x);
        fe_to_wire(o + 8, y);
    }
    out_wire[4 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
    out_wire[4 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out_wire[4 * (size_t)i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out_wire[4 * (size_t)i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

int32_t g1_ifft_run(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint64_t* out_xy) {
    int log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    NttTables tb;
    if (log_n > 0) {
        int32_t rc = ntt_get_tables(ctx, log_n, true, &tb);
        if (rc != KZG_OK) return rc;
    }
    KZG_HIP_TRY(ctx, ctx->poly[0].b.reserve(n * 36 * 4));
    KZG_HIP_TRY(ctx, ctx->msm.bases_wire.reserve(n * 64));
    int32_t* planes = ctx->poly[0].b.as<int32_t>();
    hipStream_t st = ctx->stream;
    const unsigned gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_g1fft_load, dim3(gn), dim3(256), 0, st, srs->d_points, (uint32_t)n, log_n, planes);
    for (int s = 1; s <= log_n; ++s) {
        hipLaunchKernelGGL(k_g1fft_stage, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, planes, (uint32_t)n, log_n, s, tb);
    }
    hipLaunchKernelGGL(k_g1fft_finish, dim3(gn), dim3(256), 0, st, planes, (uint32_t)n, log_n, ctx->msm.bases_wire.as<uint4>());
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(out_xy, ctx->msm.bases_wire.p, n * 64, hipMemcpyDeviceToHost, st));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return KZG_OK;
}

}  // namespace kzg
