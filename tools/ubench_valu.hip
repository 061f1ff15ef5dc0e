// Instruction-issue micro-benchmark for gfx950: measures the sustained rate of the integer / fp64
// VALU instructions a Montgomery multiplier can be built from. Output decides the limb
// representation used by the field kernels (see DESIGN.md "Integer roofline").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;

#define KERNEL32(name, ASM) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 7; \
  uint32_t r[CHAINS]; \
  for (int i = 0; i < CHAINS; ++i) r[i] = seed + i; \
  for (int it = 0; it < ITERS; ++it) { \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); \
  } \
  uint32_t s = 0; for (int i = 0; i < CHAINS; ++i) s ^= r[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s; }

#define KERNEL64(name, ASM) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 7; \
  uint64_t r[CHAINS]; \
  for (int i = 0; i < CHAINS; ++i) r[i] = seed + i; \
  for (int it = 0; it < ITERS; ++it) { \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); \
  } \
  uint64_t s = 0; for (int i = 0; i < CHAINS; ++i) s ^= r[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32)); }

#define KERNELF64(name, ASM) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
  double a = 1.0 + 1e-9 * (seed + threadIdx.x), b = 1e-3 * seed; \
  double r[CHAINS]; \
  for (int i = 0; i < CHAINS; ++i) r[i] = seed + i; \
  for (int it = 0; it < ITERS; ++it) { \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); \
  } \
  double s = 0; for (int i = 0; i < CHAINS; ++i) s += r[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s; }

KERNEL32(k_add_u32,        "v_add_u32 %0, %1, %0")
KERNEL32(k_add_co_u32,     "v_add_co_u32 %0, vcc, %1, %0")
KERNEL32(k_addc_co_u32,    "v_addc_co_u32 %0, vcc, %1, %0, vcc")
KERNEL32(k_add3_u32,       "v_add3_u32 %0, %1, %2, %0")
KERNEL32(k_mul_lo_u32,     "v_mul_lo_u32 %0, %1, %0")
KERNEL32(k_mul_hi_u32,     "v_mul_hi_u32 %0, %1, %0")
KERNEL32(k_mad_u32_u24,    "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0")
KERNEL32(k_mul_u32_u24,    "v_mul_u32_u24 %0, %1, %0")
KERNEL32(k_fma_f32,        "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_xor_b32,        "v_xor_b32 %0, %1, %0")
KERNEL32(k_alignbit,       "v_alignbit_b32 %0, %1, %0, 13")
KERNEL32(k_cndmask,        "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL64(k_mad_u64_u32,    "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32,    "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64,   "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_lshlrev_b64,    "v_lshlrev_b64 %0, 1, %0")
KERNELF64(k_fma_f64,       "v_fma_f64 %0, %1, %2, %0")
KERNELF64(k_add_f64,       "v_add_f64 %0, %1, %0")
KERNELF64(k_mul_f64,       "v_mul_f64 %0, %1, %0")

// mixed: one mad_u64_u32 + K cheap adds, to see whether the multiplier and the adder co-issue
__global__ void __launch_bounds__(256) k_mix_mad_add(uint32_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 7;
  uint64_t r[CHAINS]; uint32_t q[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { r[i] = seed + i; q[i] = seed * i; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(q[i]) : "v"(a));
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(q[i]) : "v"(b));
    }
  }
  uint64_t s = 0; for (int i = 0; i < CHAINS; ++i) s ^= r[i] + q[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int inst_per_iter; };

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu  regs/block=%d  L2=%d  mem=%zu MiB  gcn=%s\n", p.name,
         p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock, p.regsPerBlock, p.l2CacheSize,
         p.totalGlobalMem >> 20, p.gcnArchName);
  int blocks = p.multiProcessorCount * 8, threads = 256;
  uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * threads * 4));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add_co_u32", k_add_co_u32, 1}, {"v_addc_co_u32", k_addc_co_u32, 1},
    {"v_add3_u32", k_add3_u32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_alignbit_b32", k_alignbit, 1},
    {"v_cndmask_b32", k_cndmask, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
    {"v_fma_f32", k_fma_f32, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
    {"mix(mad_u64_u32+2add)", k_mix_mad_add, 3},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (auto& e : es) {
    e.k<<<blocks, threads>>>(d, 1); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) e.k<<<blocks, threads>>>(d, 2 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double lane_ops = (double)blocks * threads * ITERS * CHAINS * e.inst_per_iter;
    double wave_inst = lane_ops / 64;
    // cycles per wave-instruction per SIMD at the nominal max clock (2.4 GHz)
    double simds = p.multiProcessorCount * 4.0;
    double cyc = (ms * 1e-3 * 2.4e9) * simds / wave_inst;
    printf("%-24s %8.3f ms  %8.2f Glaneop/s  %6.2f cyc/wave-inst/SIMD@2.4GHz\n", e.name, ms, lane_ops / ms * 1e-6, cyc);
  }
  return 0;
}
