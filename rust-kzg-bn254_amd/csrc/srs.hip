// srs.hip — device-side SRS utilities.
//   k_srs_powers: synthetic SRS with a KNOWN tau, P_i = tau^i * G1 (G1 = (1, 2)), used by the tests and by
//   bench.py for the known-tau oracle  commit(p) == p(tau) * G1  (SURVEY.md §8c/§8d "SRS-S").  The reference
//   has no such routine (it loads ceremony files, prover/src/srs.rs:35-49); this is a data generator, not a
//   replacement of a reference interface.
//   k_points_device_to_wire: read a device-resident SRS back in wire format (for parity checks).
#include "engine.h"
#include "curve.h"
#include "fe_invert.h"

#include <cstdlib>

namespace kzg {

template <class F>
__device__ __forceinline__ void fe_inverse(Fe<F>& out, const Fe<F>& a) {     // a^-1 (0 -> 0): division steps, fe_invert.h (round 4; was a^(m-2))
#if !defined(KZG_INVERT_FERMAT)
    fe_inverse_safegcd(out, a);
#else
    Fe<F> acc, base = a;
    fe_set_one(acc);
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = F::P32[j];
    e[0] -= 2u;                                   // both moduli end in ...7 / ...1 with low word >= 2
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
#endif
}

// scalars: n canonical 256-bit integers k_i (device, 8 u32 each); out: device affine format
__global__ void __launch_bounds__(256)
k_srs_powers(const uint4* __restrict__ scalars_canonical, uint4* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 lo = scalars_canonical[2 * (size_t)i], hi = scalars_canonical[2 * (size_t)i + 1];
    uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    Affine g;                                     // generator (1, 2) in internal Montgomery form
    fe_set_one(g.x);
    fe_add(g.y, g.x, g.x);
    fe_norm(g.y);
    fe_canon(g.y);
    Xyzz acc;
    xyzz_set_inf(acc);
    for (int w = 7; w >= 0; --w) {
        for (int b = 31; b >= 0; --b) {
            Xyzz t;
            xyzz_dbl_impl(t, acc);
            acc = t;
            if ((k[w] >> b) & 1u) xyzz_madd<true>(acc, g, 0);
        }
    }
    uint32_t o[16];
    if (acc.inf) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0;
    } else {
        Fq zi, t, x, y;
        fe_mul(t, acc.zz, acc.zzz);
        fe_inverse(zi, t);                        // 1 / (ZZ ZZZ)
        fe_mul(t, zi, acc.zzz);                   // 1 / ZZ
        fe_mul(x, acc.x, t);
        fe_mul(t, zi, acc.zz);                    // 1 / ZZZ
        fe_mul(y, acc.y, t);
        fe_canon(x);
        fe_canon(y);
        fe_pack(o, x);
        fe_pack(o + 8, y);
    }
    out[4 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[4 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out[4 * (size_t)i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out[4 * (size_t)i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

// canonical powers tau^i (8 u32 each), lane t computes a run of `per` consecutive powers
__global__ void __launch_bounds__(256)
k_fr_powers(const uint4* __restrict__ tau_wire, uint4* __restrict__ out_canonical, uint64_t first_power, uint32_t n, uint32_t per) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t start = (uint64_t)t * per;
    if (start >= n) return;
    uint4 a = tau_wire[0], b = tau_wire[1];
    uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Fr tau, cur, base;
    fe_from_wire(tau, w32);
    // cur = tau^(first_power + start)
    fe_set_one(cur);
    base = tau;
    for (uint64_t e = first_power + start; e != 0; e >>= 1) {
        if (e & 1u) fe_mul(cur, cur, base);
        fe_sqr(base, base);
    }
    for (uint32_t k = 0; k < per && start + k < n; ++k) {
        Fr one_plain, c;
        fe_set_zero(one_plain);
        one_plain.l[0] = 1;
        fe_mul(c, cur, one_plain);                // internal -> plain integer
        fe_canon(c);
        uint32_t o[8];
        fe_pack(o, c);
        out_canonical[2 * (start + k)] = make_uint4(o[0], o[1], o[2], o[3]);
        out_canonical[2 * (start + k) + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        fe_mul(cur, cur, tau);
    }
}

// One step of the window-table precomputation: next[i] = 2^c * prev[i] (affine in, affine out).
// c doublings in XYZZ, then one Fermat inversion per point.  One-time cost at SRS upload.
__global__ void __launch_bounds__(256)
k_srs_window_step(const uint4* __restrict__ prev, uint4* __restrict__ next, size_t n, int c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p;
    uint32_t o[16];
    if (!affine_load(p, prev + 4 * i)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0;
    } else {
        Xyzz acc;
        xyzz_dbl_affine_impl(acc, p.x, p.y);
        for (int k = 1; k < c; ++k) {
            Xyzz t;
            xyzz_dbl_impl(t, acc);
            acc = t;
        }
        Fq zi, t, x, y;
        fe_mul(t, acc.zz, acc.zzz);
        fe_inverse(zi, t);
        fe_mul(t, zi, acc.zzz);
        fe_mul(x, acc.x, t);
        fe_mul(t, zi, acc.zz);
        fe_mul(y, acc.y, t);
        fe_canon(x);
        fe_canon(y);
        fe_pack(o, x);
        fe_pack(o + 8, y);
    }
    next[4 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    next[4 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    next[4 * i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    next[4 * i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

// One level of the per-bit tables: next[i] = 2 * prev[i], affine in, affine out (y^2 = x^3 + 3: lambda = 3 x^2 / (2 y),
// x3 = lambda^2 - 2 x, y3 = lambda (x - x3) - y; no point of BN254 G1 has y = 0, the group order is odd).  The inversions are
// batched: a lane owns BITS_K points (a grid stride apart, so the loads of a wave stay contiguous), keeps the running products
// of their denominators in LDS, inverts the last one (division steps, fe_invert.h: ~63 products' worth of cheap instructions; a^(m-2) before round 4) and
// walks back: 7 products per point + one inversion per BITS_K points instead of k_srs_window_step's one inversion per point.  255 levels of a 2^20-point SRS: ~0.1 s, once.
constexpr int BITS_K = 7;                               // 7 x 9 limbs x 256 lanes x 4 B = 63 KiB of LDS
__global__ void __launch_bounds__(256)
k_srs_double_batch(const uint4* __restrict__ prev, uint4* __restrict__ next, size_t n) {
    __shared__ int32_t pre[BITS_K * NL * 256];
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tl = threadIdx.x;
    Fq run;
    fe_set_one(run);
    for (int q = 0; q < BITS_K; ++q) {
        const size_t i = i0 + (size_t)q * lanes;
        Affine p;
        Fq den;
        fe_set_one(den);
        if (i < n && affine_load(p, prev + 4 * i)) { fe_dbl(den, p.y); fe_norm(den); }     // 2 y in [0, 2m)
        fe_mul(run, run, den);
#pragma unroll
        for (int j = 0; j < NL; ++j) pre[(q * NL + j) * 256 + tl] = run.l[j];
    }
    Fq inv;
    fe_inverse(inv, run);                               // 1 / (product of the lane's denominators)
    for (int q = BITS_K - 1; q >= 0; --q) {
        const size_t i = i0 + (size_t)q * lanes;
        if (i >= n) continue;                           // (its denominator was 1)
        Affine p;
        uint32_t o[16];
        if (!affine_load(p, prev + 4 * i)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] = 0;
        } else {
            Fq before, den, iq, xx, m3, lam, x3, y3, t;
            if (q == 0) fe_set_one(before);
            else {
#pragma unroll
                for (int j = 0; j < NL; ++j) before.l[j] = pre[((q - 1) * NL + j) * 256 + tl];
            }
            fe_mul(iq, inv, before);                    // 1 / (2 y)
            fe_dbl(den, p.y); fe_norm(den);
            fe_mul(inv, inv, den);
            fe_sqr(xx, p.x);
            fe_add(m3, xx, xx); fe_add(m3, m3, xx); fe_norm(m3);          // 3 x^2 in (-3m, 6m)
            fe_mul(lam, m3, iq);
            fe_sqr(x3, lam);
            fe_canon(x3);
            fe_sub(x3, x3, p.x); fe_norm(x3); fe_canon(x3);               // (-m, m) -> [0, m)
            fe_sub(x3, x3, p.x); fe_norm(x3); fe_canon(x3);
            fe_sub(t, p.x, x3); fe_norm(t);                               // (-m, m)
            fe_mul(y3, lam, t);
            fe_canon(y3);
            fe_sub(y3, y3, p.y); fe_norm(y3); fe_canon(y3);
            fe_pack(o, x3);
            fe_pack(o + 8, y3);
        }
        next[4 * i] = make_uint4(o[0], o[1], o[2], o[3]);
        next[4 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        next[4 * i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
        next[4 * i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
    }
}

// Batch decompression of gnark-format compressed G1 points (prover/src/srs.rs:51-68 -> primitives/src/helpers.rs:175-226):
// 32 big-endian bytes per point, top two bits of byte 0: 01 = infinity (rest must be zero), 10 = smaller y, 11 = larger y;
// x = remaining 254 bits mod p; y = sqrt(x^3 + 3) = (x^3 + 3)^((p+1)/4) (p = 3 mod 4), sign chosen by
// lexicographically_largest (helpers.rs:151-173: y > (p-1)/2 on the canonical integer).  The subgroup check of the
// reference is vacuous on BN254 G1 (cofactor 1).  One lane per point; status[0] = first error (0 ok, 1 bad infinity
// encoding, 2 not on curve), status[1] = its index.
//
// ARK_LE = true: the "native" format of SRS::parallel_read_g1_points_native(is_native = true) (prover/src/srs.rs:205-251 ->
// primitives/src/traits.rs:34-36, G1Affine::deserialize_compressed of ark-serialize 0.5, restated -- the reference holds no file in this
// format): x as 32 LITTLE-endian bytes, flags in the top two bits of the LAST byte: 0x80 = y is the larger root (SWFlags::YIsNegative),
// 0x40 = the point at infinity, both = rejected; x must be canonical (< p), else the element does not deserialise (status 1).
template <bool ARK_LE>
__global__ void __launch_bounds__(256)
k_srs_decompress(const uint8_t* __restrict__ bytes, uint4* __restrict__ out, uint32_t n, uint32_t* __restrict__ status) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = bytes + (size_t)i * 32;
    const uint32_t flag = (ARK_LE ? b[31] : b[0]) & 0xC0u;
    constexpr uint32_t FLAG_SMALLEST = ARK_LE ? 0x00u : 0x80u, FLAG_LARGEST = ARK_LE ? 0x80u : 0xC0u;
    uint32_t o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0;
    uint32_t err = 0;
    uint32_t w32[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                        // little-endian word k = big-endian bytes 28-4k .. 31-4k (gnark) / bytes 4k .. 4k+3 (ark)
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t byte = ARK_LE ? b[4 * k + 3 - t] : b[28 - 4 * k + t];
            if (k == 7 && t == 0) byte &= 0x3Fu;
            w = (w << 8) | byte;
        }
        w32[k] = w;
    }
    bool x_canonical = false;                            // x < p ?
    for (int k = 7; k >= 0; --k) {
        if (w32[k] != FqParams::P32[k]) { x_canonical = w32[k] < FqParams::P32[k]; break; }
    }
    if (ARK_LE && (flag == 0xC0u || !x_canonical)) {
        err = 1;                                         // ark-serialize: UnexpectedFlags / a field element >= the modulus
    } else if (flag == 0x40u) {
        uint32_t rest = 0;
        for (int k = 0; k < 8; ++k) rest |= w32[k];
        if (rest && !ARK_LE) err = 1;                    // "point at infinity not coded properly for g1" (ark returns the identity whatever x)
    } else {
        Fq x, y2, y, t, kin, b3;
        fe_unpack(x, w32);
#pragma unroll
        for (int j = 0; j < NL; ++j) { kin.l[j] = (int32_t)FqParams::K_PLAIN_IN[j]; b3.l[j] = (int32_t)FqParams::B3[j]; }
        fe_mul(x, x, kin);                               // Fq::from_be_bytes_mod_order -> internal form, (-m, 2m)
        fe_sqr(t, x);
        fe_mul(y2, t, x);
        fe_add(y2, y2, b3);
        fe_norm(y2);                                     // x^3 + 3, |.| < 4m
        // y = y2^((p+1)/4), square-and-multiply from the least significant bit
        {
            Fq acc, base = y2;
            fe_set_one(acc);
            for (int w = 0; w < 8; ++w) {
                uint32_t bits = FqParams::SQRT_EXP32[w];
                for (int bb = 0; bb < 32; ++bb) {
                    if (w == 7 && bb >= 28) break;       // (p+1)/4 < 2^252
                    if (bits & 1u) fe_mul(acc, acc, base);
                    fe_sqr(base, base);
                    bits >>= 1;
                }
            }
            y = acc;
        }
        fe_sqr(t, y);
        fe_sub(t, t, y2);
        fe_reduce(t);
        if (!fe_is_zero_mod(t)) {
            err = 2;                                     // "compressed g1 point not on curve"
        } else {
            // canonical integers of x (internal form) and of y (plain) for the sign rule
            Fq one_plain, yc;
            fe_set_zero(one_plain);
            one_plain.l[0] = 1;
            fe_mul(yc, y, one_plain);
            fe_canon(yc);
            uint32_t yw[8];
            fe_pack(yw, yc);
            bool ge = true;                              // yc >= (p+1)/2 ?
            for (int k = 7; k >= 0; --k) {
                if (yw[k] != FqParams::HALF_UP32[k]) { ge = yw[k] > FqParams::HALF_UP32[k]; break; }
            }
            const bool negate = ge ? (flag == FLAG_SMALLEST) : (flag == FLAG_LARGEST);
            Fq ys;
            fe_canon(y);                                 // [0, m) before the sign flip so that -y stays inside (-m, 2m)
            fe_cneg(ys, y, negate ? 1u : 0u);
            fe_norm(ys);
            fe_canon(x);
            fe_canon(ys);
            fe_pack(o, x);
            fe_pack(o + 8, ys);
        }
    }
    if (err) {
        if (atomicCAS(&status[0], 0u, err) == 0u) status[1] = i;
    }
    out[4 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[4 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out[4 * (size_t)i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out[4 * (size_t)i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

// device affine format -> wire (x || y, radix 2^256)
__global__ void __launch_bounds__(256)
k_points_device_to_wire(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p;
    bool ok = affine_load(p, in + 4 * i);
    uint32_t o[16];
    if (!ok) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0;
    } else {
        fe_to_wire(o, p.x);
        fe_to_wire(o + 8, p.y);
    }
    out[4 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[4 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out[4 * i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out[4 * i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

int32_t srs_generate(kzg_ctx* ctx, const uint64_t tau[4], uint64_t first_power, size_t n, uint4* d_points) {
    if (n == 0) return KZG_OK;
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * 32 + 64));
    KZG_HIP_TRY(ctx, ctx->poly[0].small.reserve(4096));
    uint4* d_tau = ctx->poly[0].small.as<uint4>();
    KZG_HIP_TRY(ctx, hipMemcpyAsync(d_tau, tau, 32, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t per = 64;
    uint32_t lanes = (uint32_t)((n + per - 1) / per);
    hipLaunchKernelGGL(k_fr_powers, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, d_tau, ctx->poly[0].a.as<uint4>(), first_power, (uint32_t)n, per);
    hipLaunchKernelGGL(k_srs_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->poly[0].a.as<uint4>(), d_points, (uint32_t)n);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

// Window tables T_w = 2^(c w) * SRS for w < W, contiguous after the SRS itself.  Spends HBM capacity (W x 64 B per
// point: 0.94 GiB for 2^20 points at c = 17) to turn the MSM into W n mixed adds into a single bucket set.
int32_t srs_precompute(kzg_ctx* ctx, kzg_srs* srs) {
    if (opt_no_precompute()) return KZG_OK;
    const size_t n = srs->n;
    if (n < 128) return KZG_OK;
    int lg = 0;
    while ((n >> (lg + 1)) != 0) ++lg;
    // window bits, measured with the equal-split accumulate and the two-level sort (tools/sweep_shard_c.py, tools/time_small.py):
    // the bucket reduction costs about the same ~0.17 ms for 2^12..2^16 buckets (one wave per 64 buckets, all resident), so the
    // window count decides: c = 17 (15 windows) from 2^18 points, c = 15 (17 windows) for 2^14..2^17 (2^17 pairs, three in
    // flight: 0.248 ms at c = 15, 0.284 at c = 17, 0.294 at c = 13), c = 13 below (narrow windows on a small SRS mean few, heavy
    // buckets, which the reduction sums one after the other: 2^12 points at c = 8 took 2.6 ms, 0.35 ms at c = 13).  Above 2^20
    // points the 24-bit index of the two-level sort no longer holds W * n: those MSMs run as chunks of 2^20 pairs with compact
    // indices (msm.hip msm_run).
    int c = 13;
    if (lg >= 14) c = 15;
    if (lg >= 18) c = 17;
    // tables T_w = 2^(cw) * SRS, w < W, as one allocation (table 0 = a copy of the SRS); nullptr when they do not fit
    auto build = [&](int cw, uint4** out, int* out_W) -> int32_t {
        *out = nullptr;
        const int W = (255 + cw - 1) / cw;
        const size_t bytes = (size_t)W * n * 64;
        if (bytes > ((size_t)48 << 30) || (size_t)W * n >= ((size_t)1 << 31)) return KZG_OK;
        uint4* table = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&table), bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return KZG_OK; }      // not enough memory: stay without
        e = hipMemcpyAsync(table, srs->d_points, n * 64, hipMemcpyDeviceToDevice, ctx->stream);
        for (int w = 1; w < W && e == hipSuccess; ++w) {
            hipLaunchKernelGGL(k_srs_window_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               table + 4 * (size_t)(w - 1) * n, table + 4 * (size_t)w * n, n, cw);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { (void)hipFree(table); return set_error(ctx, e, "building the SRS window tables"); }   // (the table leaked on these paths)
        *out = table; *out_W = W;
        return KZG_OK;
    };
    uint4* table = nullptr;
    int W = 0;
    { int32_t rc = build(c, &table, &W); if (rc != KZG_OK) return rc; }
    if (!table) return KZG_OK;                                                // generic mode
    // A second, narrower table set for SMALL MSMs over this SRS: every launch pays for all 2^(c-1) buckets (scan, empty-bucket
    // walks, 2^(c-7) reduction waves), which a 2^11..2^13-coefficient commitment does not amortise -- 0.36..0.42 ms on the c = 17
    // tables of a 2^20-point SRS, 0.27..0.31 ms on c = 15 ones (tools/time_commit_sizes.py).  HBM is plentiful (another 1.06 GiB
    // at 2^20 points), so both are kept and srs_bases() picks per launch.
    if (c > SRS_SMALL_C) {
        uint4* t2 = nullptr; int W2 = 0;
        int32_t rc = build(SRS_SMALL_C, &t2, &W2);
        if (rc != KZG_OK) { (void)hipFree(table); return rc; }
        if (t2) { srs->d_small = t2; srs->small_c = SRS_SMALL_C; srs->small_W = W2; }
    }
    KZG_HIP_TRY(ctx, hipFree(srs->d_points));
    srs->d_points = table;
    srs->pre_c = c;
    srs->pre_W = W;
    return srs_build_bit_tables(ctx, srs, false);
}

// Per-bit tables Bit_j[i] = 2^j P_i, j < 255 (msm_kernels.h "NAF mode"): 255 x 64 B per point -- 15.9 GiB for a 2^20-point SRS, of
// the 288 GB this GPU has -- so that MSMs of >= SRS_NAF_MIN pairs recode their scalars in width-w NAF: 254 / (w + 1) mixed additions
// per scalar instead of 255 / c into the same 2^(c-1) buckets (13.4 instead of 15 at 2^16 buckets; measured in DESIGN.md section 4c).
// Built for SRS of 2^11 .. 2^22 points when they fit beside a quarter of the free memory; KZG_NO_NAF=1: never.
// force: also for an SRS below SRS_NAF_MIN points (the batched commitments build them on first use: 16 KiB per point)
int32_t srs_build_bit_tables(kzg_ctx* ctx, kzg_srs* srs, bool force) {
    if (opt_no_naf()) return KZG_OK;
    const size_t n = srs->n;
    std::lock_guard<std::mutex> lazy(srs->lazy_mu);           // two contexts asking at once: the second finds the tables built
    if (n == 0 || (n < SRS_NAF_MIN && !force) || n > ((size_t)1 << 22) || srs->d_bits) return KZG_OK;
    const size_t bytes = (size_t)255 * n * 64;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return KZG_OK; }
    if (bytes + (bytes >> 2) + ((size_t)8 << 30) > free_b) return KZG_OK;               // keep room for workspaces and other SRS
    uint4* bits = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&bits), bytes) != hipSuccess) { (void)hipGetLastError(); return KZG_OK; }
    hipError_t e = hipMemcpyAsync(bits, srs->d_points, n * 64, hipMemcpyDeviceToDevice, ctx->stream);
    const size_t lanes = (n + BITS_K - 1) / BITS_K;
    const unsigned blocks = (unsigned)((lanes + 255) / 256);
    for (int j = 1; j < 255 && e == hipSuccess; ++j)
        hipLaunchKernelGGL(k_srs_double_batch, dim3(blocks), dim3(256), 0, ctx->stream, bits + 4 * (size_t)(j - 1) * n, bits + 4 * (size_t)j * n, n);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(bits); return set_error(ctx, e, "building the per-bit SRS tables"); }
    srs->d_bits = bits;
    return KZG_OK;
}

// compressed big-endian points (host bytes) -> d_points (device format); *err_kind / *err_index describe the first bad point
int32_t srs_decompress(kzg_ctx* ctx, const uint8_t* bytes, size_t n, uint4* d_points, uint32_t* err_kind, uint32_t* err_index, bool ark_le) {
    *err_kind = 0;
    *err_index = 0;
    if (n == 0) return KZG_OK;
    KZG_HIP_TRY(ctx, ctx->poly[0].c.reserve(n * 32));
    KZG_HIP_TRY(ctx, ctx->poly[0].small.reserve(4096));
    uint32_t* d_status = ctx->poly[0].small.as<uint32_t>();
    KZG_HIP_TRY(ctx, hipMemsetAsync(d_status, 0, 8, ctx->stream));
    KZG_HIP_TRY(ctx, hipMemcpyAsync(ctx->poly[0].c.p, bytes, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (ark_le)
        hipLaunchKernelGGL(k_srs_decompress<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->poly[0].c.as<uint8_t>(), d_points,
                           (uint32_t)n, d_status);
    else
        hipLaunchKernelGGL(k_srs_decompress<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->poly[0].c.as<uint8_t>(), d_points,
                           (uint32_t)n, d_status);
    KZG_HIP_TRY(ctx, hipGetLastError());
    uint32_t st[2] = {0, 0};
    KZG_HIP_TRY(ctx, hipMemcpyAsync(st, d_status, 8, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *err_kind = st[0];
    *err_index = st[1];
    return KZG_OK;
}

int32_t srs_download(kzg_ctx* ctx, const uint4* d_points, size_t n, uint64_t* out_xy) {
    if (n == 0) return KZG_OK;
    KZG_HIP_TRY(ctx, ctx->msm.bases_wire.reserve(n * 64));
    hipLaunchKernelGGL(k_points_device_to_wire, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_points,
                       ctx->msm.bases_wire.as<uint4>(), n);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipMemcpyAsync(out_xy, ctx->msm.bases_wire.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

}  // namespace kzg
