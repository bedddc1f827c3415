// ubench.hip — measurement aid behind kzg_ctx_measure_valu_rates: the issue rate of the instruction classes the MSM accumulate
// kernel is made of, measured on THIS device at that kernel's occupancy (three waves per SIMD), so that bench.py prices the kernel's
// instruction stream with the box's own rates instead of constants from another box (chips differ by several percent in the clock
// they hold under load: MI355X_MICROARCH.md "DVFS give-back").  Eight independent chains per lane, 4096 x 16 instructions per wave.
#include "engine.h"

namespace kzg {

constexpr int UB_ITERS = 16384;      // ~2 ms per kernel: long enough for the clock to settle

// the eight chains are ONE asm statement: hipcc pads every asm statement with an s_nop, which would otherwise be measured along
#define KZG_UB_ASM8(I)                                                                                         \
    I("%0") "\n\t" I("%1") "\n\t" I("%2") "\n\t" I("%3") "\n\t" I("%4") "\n\t" I("%5") "\n\t" I("%6") "\n\t" I("%7")
#define KZG_UB_KERNEL(NAME, T, I, ...)                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                \
        T a0 = (T)(seed + threadIdx.x), a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;                     \
        T a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                              \
        int b = (int)(seed * 2654435761u + threadIdx.x), c = (int)(seed ^ 0x9e3779b9u);                       \
        for (int i = 0; i < UB_ITERS; ++i) {                                                                   \
            asm volatile(KZG_UB_ASM8(I) "\n\t" KZG_UB_ASM8(I)                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                         : "v"(b), "v"(c) : __VA_ARGS__);                                                      \
        }                                                                                                      \
        const long long x = (long long)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);                        \
    }
#define KZG_UB_I_MAD(R) "v_mad_i64_i32 " R ", vcc, %8, %9, " R
#define KZG_UB_I_MULLO(R) "v_mul_lo_u32 " R ", %8, " R
#define KZG_UB_I_ASHR(R) "v_ashrrev_i64 " R ", 29, " R
#define KZG_UB_I_AND(R) "v_and_b32 " R ", %8, " R
#define KZG_UB_I_SUB(R) "v_sub_u32 " R ", %8, " R
#define KZG_UB_I_NOP(R) "s_nop 0"

KZG_UB_KERNEL(k_ub_mad_i64_i32, long long, KZG_UB_I_MAD, "vcc")
KZG_UB_KERNEL(k_ub_mul_lo_u32, int, KZG_UB_I_MULLO, "memory")
KZG_UB_KERNEL(k_ub_ashr_i64, long long, KZG_UB_I_ASHR, "memory")
KZG_UB_KERNEL(k_ub_and_b32, int, KZG_UB_I_AND, "memory")
KZG_UB_KERNEL(k_ub_sub_u32, int, KZG_UB_I_SUB, "memory")
KZG_UB_KERNEL(k_ub_nop, int, KZG_UB_I_NOP, "memory")

}  // namespace kzg

using namespace kzg;

extern "C" int32_t kzg_ctx_measure_valu_rates(kzg_ctx* ctx, int32_t waves_per_simd, double out_ns[6]) {
    if (!ctx || !out_ns || waves_per_simd < 1 || waves_per_simd > 8) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    KZG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const unsigned blocks = (unsigned)cus * (unsigned)waves_per_simd;          // 256 threads = one wave per SIMD of a CU
    uint32_t* d = nullptr;
    KZG_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), (size_t)blocks * 256 * 4));
    hipEvent_t t0, t1;
    KZG_HIP_TRY(ctx, hipEventCreate(&t0));
    KZG_HIP_TRY(ctx, hipEventCreate(&t1));
    typedef void (*kern_t)(uint32_t*, uint32_t);
    const kern_t ks[6] = {k_ub_mad_i64_i32, k_ub_mul_lo_u32, k_ub_ashr_i64, k_ub_and_b32, k_ub_sub_u32, k_ub_nop};
    int32_t rc = KZG_OK;
    for (int i = 0; i < 6 && rc == KZG_OK; ++i) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {                                           // the first repetitions warm the clock up
            (void)hipEventRecord(t0, ctx->stream);
            hipLaunchKernelGGL(ks[i], dim3(blocks), dim3(256), 0, ctx->stream, d, 12345u + (uint32_t)r);
            (void)hipEventRecord(t1, ctx->stream);
            if (hipEventSynchronize(t1) != hipSuccess) { rc = KZG_ERR_DEVICE; break; }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, t0, t1);
            if (r > 1 && ms < best) best = ms;
        }
        out_ns[i] = (double)best * 1e6 / ((double)UB_ITERS * 16.0 * (double)waves_per_simd);   // ns per wave-instruction per SIMD
    }
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    (void)hipFree(d);
    return rc;
}
