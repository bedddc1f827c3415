// Device check of the lane-quad point arithmetic (csrc/curve_quad.h) against the one-lane formulas (csrc/curve.h): quad_add and quad_madd on
// random field values (the formulas are rational maps: the operands need not be curve points), with the identity, doubling and
// cancellation cases mixed into every wave; round 4: quad_dbl against xyzz_dbl.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 quad_check.hip -o quad_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../rust-kzg-bn254_amd/csrc/curve_quad.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// a, b: 32 wire words per case (X || Y || ZZ || ZZZ);  kind = case % 8: 1 b = a, 2 b = -a, 3 a = identity, 4 b = identity, 5 both
__global__ void __launch_bounds__(64) k_check(const uint32_t* __restrict__ a_w, const uint32_t* __restrict__ b_w, uint32_t n, int madd,
                                              uint32_t* __restrict__ out_quad, uint32_t* __restrict__ out_ref) {
    const uint32_t lane = threadIdx.x, q = lane & 3u, cs = blockIdx.x * 16 + (lane >> 2);
    if (cs >= n) return;
    const uint32_t kind = cs % 8;
    QuadXyzz a, b;
    fe_from_wire(a.c, a_w + (size_t)cs * 32 + q * 8);
    fe_from_wire(b.c, b_w + (size_t)cs * 32 + q * 8);
    a.inf = false; b.inf = false;
    if (kind == 1 || kind == 2) b = a;
    if (kind == 2 && q == 1) { fe_neg(b.c, a.c); fe_norm(b.c); }
    if (kind == 3 || kind == 5) quad_set_inf(a);
    if (!madd && (kind == 4 || kind == 5)) quad_set_inf(b);
    Xyzz A, B, R;
    quad_gather(A, a);
    quad_gather(B, b);
    QuadXyzz r;
    if (madd == 2) {                           // doubling (round 4: quad_dbl); kind 6 / 7: X at the wide end of its range (3 x in (-3m, 6m))
        if ((kind == 6 || kind == 7) && q == 0) { Fq t; fe_add(t, a.c, a.c); fe_add(t, t, a.c); fe_norm(t); a.c = t; }
        quad_gather(A, a);
        quad_dbl_any(r, a, q);
        xyzz_dbl(R, A);
    } else if (!madd) {
        quad_add(r, a, b, q);
        xyzz_add<false>(R, A, B);
    } else {
        // the affine point (x, y) = (X, Y) of b; kind 1 / 2: a becomes (x, +-y, 1, 1) so that the mixed addition doubles / cancels
        Fq c, xs, ys;
        fe_from_wire(xs, b_w + (size_t)cs * 32);
        fe_from_wire(ys, b_w + (size_t)cs * 32 + 8);
        fe_canon(xs); fe_canon(ys);
        c = (q & 1u) ? ys : xs;
        const uint32_t neg = (cs >> 3) & 1u;
        if (kind == 1 || kind == 2) {
            Fq one; fe_set_one(one);
            Fq y1; fe_cneg(y1, ys, neg ^ (kind == 2 ? 1u : 0u)); fe_norm(y1);
            a.c = q == 0 ? xs : q == 1 ? y1 : one;
            a.inf = false;
            quad_gather(A, a);
        }
        quad_madd(r, a, c, neg, q);
        Affine P; P.x = xs; P.y = ys;
        R = A;
        xyzz_madd<false>(R, P, neg);
    }
    quad_store_wire(out_quad, cs, r, q);
    if (q == 0) xyzz_to_wire(out_ref + (size_t)cs * 32, R);
}

int main() {
    const uint32_t n = 4096;
    std::vector<uint32_t> a((size_t)n * 32), b((size_t)n * 32);
    uint64_t s = 88172645463325252ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (size_t i = 0; i < a.size(); ++i) { a[i] = next(); b[i] = next(); if (i % 8 == 7) { a[i] &= 0x1FFFFFFFu; b[i] &= 0x1FFFFFFFu; } }
    uint32_t *d_a, *d_b, *d_q, *d_r;
    CHECK(hipMalloc(&d_a, a.size() * 4)); CHECK(hipMalloc(&d_b, a.size() * 4)); CHECK(hipMalloc(&d_q, a.size() * 4)); CHECK(hipMalloc(&d_r, a.size() * 4));
    CHECK(hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, b.data(), a.size() * 4, hipMemcpyHostToDevice));
    int bad_total = 0;
    for (int madd = 0; madd < 3; ++madd) {
        CHECK(hipMemset(d_q, 0xEE, a.size() * 4)); CHECK(hipMemset(d_r, 0xDD, a.size() * 4));
        hipLaunchKernelGGL(k_check, dim3(n / 16), dim3(64), 0, 0, d_a, d_b, n, madd, d_q, d_r);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> hq(a.size()), hr(a.size());
        CHECK(hipMemcpy(hq.data(), d_q, a.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hr.data(), d_r, a.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0, bad_kind[8] = {}, bad_coord[4] = {};
        for (uint32_t i = 0; i < n; ++i) {
            bool any = false;
            for (int c = 0; c < 4; ++c)
                if (memcmp(&hq[(size_t)i * 32 + c * 8], &hr[(size_t)i * 32 + c * 8], 32) != 0) { ++bad_coord[c]; any = true; }
            if (any) { ++bad; ++bad_kind[i % 8]; }
        }
        printf("%s: %d of %u cases differ; by kind:", madd == 2 ? "quad_dbl" : madd ? "quad_madd" : "quad_add", bad, n);
        for (int k = 0; k < 8; ++k) printf(" %d", bad_kind[k]);
        printf("; by coordinate X Y ZZ ZZZ: %d %d %d %d\n", bad_coord[0], bad_coord[1], bad_coord[2], bad_coord[3]);
        bad_total += bad;
    }
    printf(bad_total ? "FAIL\n" : "OK\n");
    return bad_total ? 1 : 0;
}
