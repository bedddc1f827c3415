// Micro-benchmark: per-instruction VALU throughput on gfx950 for the instructions a
// 254-bit modular multiply can be built from.  Decides the limb representation
// (8x u32 limbs on v_mad_u64_u32 vs 52-bit limbs on v_fma_f64).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;   // loop trips
constexpr int UNROLL = 16;    // instrs per trip (8 independent chains x 2)

#define KERNEL32(NAME, ASM, ...)                                                          \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                      \
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;    \
  uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;       \
  uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                  \
  for (int i = 0; i < ITERS; ++i) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                              \
      asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : __VA_ARGS__);                               \
    }                                                                                     \
  }                                                                                       \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
}

#define KERNEL64(NAME, ASM, ...)                                                          \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                      \
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;    \
  uint64_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;       \
  uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                  \
  for (int i = 0; i < ITERS; ++i) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                              \
      asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : __VA_ARGS__);                               \
      asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : __VA_ARGS__);                               \
    }                                                                                     \
  }                                                                                       \
  uint64_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                     \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);         \
}

#define KERNELF64(NAME, ASM)                                                              \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                      \
  double a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;      \
  double a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;         \
  double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-3 * seed;                                   \
  for (int i = 0; i < ITERS; ++i) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                              \
      asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                                      \
      asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                                      \
    }                                                                                     \
  }                                                                                       \
  double x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                       \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(long long)x;                    \
}

KERNEL32(k_fma_f32,        "v_fma_f32 %0, %1, %2, %0", "memory")
KERNEL32(k_add_u32,        "v_add_u32 %0, %1, %0", "memory")
KERNEL32(k_add_co_u32,     "v_add_co_u32 %0, vcc, %1, %0", "vcc")
KERNEL32(k_addc_co_u32,    "v_addc_co_u32 %0, vcc, %1, %0, vcc", "vcc")
KERNEL32(k_add3_u32,       "v_add3_u32 %0, %1, %2, %0", "memory")
KERNEL32(k_mul_lo_u32,     "v_mul_lo_u32 %0, %1, %0", "memory")
KERNEL32(k_mul_hi_u32,     "v_mul_hi_u32 %0, %1, %0", "memory")
KERNEL32(k_mad_u32_u24,    "v_mad_u32_u24 %0, %1, %2, %0", "memory")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0", "memory")
KERNEL32(k_mad_u32_u16,    "v_mad_u32_u16 %0, %1, %2, %0", "memory")
KERNEL32(k_alignbit,       "v_alignbit_b32 %0, %1, %0, 7", "memory")
KERNEL32(k_lshl_or,        "v_lshl_or_b32 %0, %1, 3, %0", "memory")
KERNEL64(k_mad_u64_u32,    "v_mad_u64_u32 %0, vcc, %1, %2, %0", "vcc")
KERNEL64(k_mad_u64_u32_s,  "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "s20", "s21")
KERNEL64(k_lshl_add_u64,   "v_lshl_add_u64 %0, %0, 0, %0", "memory")
KERNELF64(k_fma_f64,       "v_fma_f64 %0, %1, %2, %0")
KERNELF64(k_add_f64,       "v_add_f64 %0, %1, %0")
KERNELF64(k_mul_f64,       "v_mul_f64 %0, %1, %0")

// single DEPENDENT chain (what a product-scanning column accumulator looks like)
#define KERNEL64DEP(NAME, ASM, ...)                                                       \
__global__ void NAME(uint32_t* out, uint32_t seed) {                                      \
  uint64_t a0 = seed + threadIdx.x;                                                       \
  uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                  \
  for (int i = 0; i < ITERS; ++i) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                  \
      asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : __VA_ARGS__);                        \
    }                                                                                     \
  }                                                                                       \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32);       \
}
KERNEL64DEP(k_mad_u64_dep,  "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "s20", "s21")
KERNEL64DEP(k_mad_i64_dep,  "v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "s20", "s21")
KERNEL64(k_mad_i64_i32,     "v_mad_i64_i32 %0, s[20:21], %1, %2, %0", "s20", "s21")
KERNEL64(k_ashr_i64,        "v_ashrrev_i64 %0, 29, %0", "memory")
KERNEL32(k_and_b32,         "v_and_b32 %0, %1, %0", "memory")
KERNEL32(k_sub_u32,         "v_sub_u32 %0, %1, %0", "memory")
KERNEL32(k_ashr_i32,        "v_ashrrev_i32 %0, 29, %0", "memory")
KERNEL32(k_mov_b32,         "v_mov_b32 %0, %1", "memory")
KERNEL32(k_cndmask,         "v_cndmask_b32 %0, %1, %0, vcc", "memory")

// two interleaved dependent chains
__global__ void k_mad_2chains(uint32_t* out, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1;
  uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 2; ++u) {
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "s20", "s21");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c) : "s20", "s21");
    }
  }
  uint64_t x = a0 ^ a1;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}

// mixed: one mad_u64_u32 followed by two full-rate adds (the CIOS inner step shape)
__global__ void k_mix_mad_2add(uint32_t* out, uint32_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  uint32_t x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
  uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_add_co_u32 %1, vcc, %2, %1\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(a0), "+v"(x0) : "v"(b), "v"(c) : "vcc", "s20", "s21");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_add_co_u32 %1, vcc, %2, %1\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(a1), "+v"(x1) : "v"(b), "v"(c) : "vcc", "s20", "s21");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_add_co_u32 %1, vcc, %2, %1\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(a2), "+v"(x2) : "v"(b), "v"(c) : "vcc", "s20", "s21");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_add_co_u32 %1, vcc, %2, %1\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(a3), "+v"(x3) : "v"(b), "v"(c) : "vcc", "s20", "s21");
    }
  }
  uint64_t x = a0 ^ a1 ^ a2 ^ a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32) ^ x0 ^ x1 ^ x2 ^ x3;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int instrs_per_slot; };

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  std::vector<Entry> es = {
    {"v_fma_f32", k_fma_f32, 1}, {"v_add_u32", k_add_u32, 1}, {"v_add_co_u32", k_add_co_u32, 1},
    {"v_addc_co_u32", k_addc_co_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_mad_u32_u16", k_mad_u32_u16, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_lshl_or_b32", k_lshl_or, 1},
    {"v_mad_u64_u32(vcc)", k_mad_u64_u32, 1}, {"v_mad_u64_u32(sgpr)", k_mad_u64_u32_s, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
    {"mad_u64+add_co+addc", k_mix_mad_2add, 3},
    {"v_mad_u64_u32 1 dep chain", k_mad_u64_dep, 1}, {"v_mad_i64_i32 1 dep chain", k_mad_i64_dep, 1},
    {"v_mad_u64_u32 2 dep chains", k_mad_2chains, 1}, {"v_mad_i64_i32 (8 chains)", k_mad_i64_i32, 1},
    {"v_ashrrev_i64", k_ashr_i64, 1}, {"v_and_b32", k_and_b32, 1}, {"v_sub_u32", k_sub_u32, 1},
    {"v_ashrrev_i32", k_ashr_i32, 1}, {"v_mov_b32", k_mov_b32, 1}, {"v_cndmask_b32", k_cndmask, 1},
  };
  uint32_t* d; 
  for (int wavesPerSimd : {1, 2, 3, 4, 8}) {
    int threads = 256;                       // 4 waves = 1 per SIMD
    int blocks = cus * wavesPerSimd;         // wavesPerSimd blocks per CU
    CHECK(hipMalloc(&d, (size_t)blocks * threads * 4));
    printf("--- %d wave(s) per SIMD (grid %d x %d)\n", wavesPerSimd, blocks, threads);
    for (auto& e : es) {
      hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(t0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
        CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
        float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms;
      }
      // wave-instructions per SIMD
      int slots = (e.instrs_per_slot == 3) ? UNROLL : UNROLL;
      double winstr = (double)ITERS * slots * e.instrs_per_slot * wavesPerSimd;
      double ns_per = best * 1e6 / winstr;
      printf("%-22s %8.3f ms  %7.3f ns/wave-instr/SIMD  (= %6.2f cyc @2.4GHz)  chip %8.2f Gwave-instr/s\n",
             e.name, best, ns_per, ns_per * 2.4, winstr * cus * 4 / (best * 1e-3) / 1e9);
    }
    CHECK(hipFree(d));
  }
  return 0;
}
