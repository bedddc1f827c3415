// poly_common.h -- helpers shared by the polynomial kernels (poly.hip: whole-domain proofs) and the evaluation-slice kernels of the
// Lagrange-sharded proofs (lagrange.hip): limb-plane loads / stores, domain elements from the two-level twiddle tables, wire <-> internal
// conversion, the workgroup sum, and host-side Fr arithmetic on wire words for the handful of scalars a proof needs on the host.
#pragma once
#include "engine.h"
#include "field29.h"
#include "host_fr.h"

#include <cstring>

namespace kzg {

constexpr int POLY_THREADS = 256;
constexpr uint32_t NO_INDEX = 0xFFFFFFFFu;

__device__ __forceinline__ void pl_load(Fr& v, const int32_t* __restrict__ planes, size_t stride, size_t i) {
#pragma unroll
    for (int j = 0; j < NL; ++j) v.l[j] = planes[(size_t)j * stride + i];
}
__device__ __forceinline__ void pl_store(int32_t* __restrict__ planes, size_t stride, size_t i, const Fr& v) {
#pragma unroll
    for (int j = 0; j < NL; ++j) planes[(size_t)j * stride + i] = v.l[j];
}
__device__ __forceinline__ void domain_elem(Fr& w, const NttTables& tb, uint32_t E) {   // w^E, result in (-m, 2m)
    pl_load(w, tb.lo, tb.lo_len, E & (tb.lo_len - 1));
    uint32_t eh = E >> tb.lo_bits;
    if (eh != 0) {
        Fr h;
        pl_load(h, tb.hi, tb.hi_len, eh);
        fe_mul(w, w, h);
    }
}
__device__ __forceinline__ void wire_load(Fr& v, const uint4* __restrict__ src, size_t i) {   // wire -> internal, (-m, 2m)
    uint4 a = src[2 * i], b = src[2 * i + 1];
    uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fe_from_wire(v, w32);
}
__device__ __forceinline__ void wire_store(uint4* __restrict__ dst, size_t i, const Fr& v) {  // internal (|v| < 169 m) -> wire
    uint32_t w32[8];
    fe_to_wire(w32, v);
    dst[2 * i] = make_uint4(w32[0], w32[1], w32[2], w32[3]);
    dst[2 * i + 1] = make_uint4(w32[4], w32[5], w32[6], w32[7]);
}
// block-wide sum of one Fr per thread (values in (-m, 2m)); result (reduced) valid in thread 0
__device__ __forceinline__ void block_sum(Fr& v, int32_t* lds /* NL * POLY_THREADS */) {
    const int t = threadIdx.x;
    int level = 0;
    for (int d = POLY_THREADS / 2; d >= 1; d >>= 1, ++level) {
#pragma unroll
        for (int j = 0; j < NL; ++j) lds[j * POLY_THREADS + t] = v.l[j];
        __syncthreads();
        if (t < d) {
            Fr u;
#pragma unroll
            for (int j = 0; j < NL; ++j) u.l[j] = lds[j * POLY_THREADS + t + d];
            fe_add(v, v, u);
            fe_norm(v);
            if (level == 3) fe_reduce(v);        // 16 terms so far: |v| < 32 m -> back to (-m, 2m)
        }
        __syncthreads();
    }
    if (t == 0) fe_reduce(v);
}

}  // namespace kzg
