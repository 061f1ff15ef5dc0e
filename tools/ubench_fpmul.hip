// Field-multiplication throughput micro-benchmark (gfx950): saturated CIOS vs unsaturated product-scanning variants,
// at different occupancies. Each thread runs a dependent chain x <- x*y (+ y <- y*x) so nothing can be hoisted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define GMSM_INLINE_MUL 1
#include "../gnark-crypto_amd/csrc/gmsm_fieldu.h"
using namespace gmsm;
using P = bn254_fp_params;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;

// variant: two interleaved column accumulators (even / odd partial products) to double the ILP of the mad chain
template <class PP>
__device__ __forceinline__ FpU<PP> fpu_mul2(const FpU<PP> &a, const FpU<PP> &b) {
    constexpr int L = PP::UL, W = PP::UW;
    constexpr uint32_t MASK = FpU<PP>::MASK;
    uint32_t m[L];
    FpU<PP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc2 += (uint64_t)m[i] * PP::UQ[k - i];
        acc += acc2;
        m[k] = ((uint32_t)acc * PP::UQINV) & MASK;
        acc += (uint64_t)m[k] * PP::UQ[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) acc2 += (uint64_t)m[i] * PP::UQ[k - i];
        acc += acc2;
        r.l[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

// variant: single accumulator chain forced with inline asm (no v_lshl_add_u64 merges)
__device__ __forceinline__ void mad_vv(uint64_t &acc, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mad_vs(uint64_t &acc, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc");
}
template <class PP>
__device__ __forceinline__ FpU<PP> fpu_mul_asm(const FpU<PP> &a, const FpU<PP> &b) {
    constexpr int L = PP::UL, W = PP::UW;
    constexpr uint32_t MASK = FpU<PP>::MASK;
    uint32_t m[L];
    FpU<PP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) mad_vs(acc, m[i], PP::UQ[k - i]);
        m[k] = ((uint32_t)acc * PP::UQINV) & MASK;
        mad_vs(acc, m[k], PP::UQ[0]);
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - L + 1; i < L; ++i) mad_vs(acc, m[i], PP::UQ[k - i]);
        r.l[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

template <int MODE, int MINW>
__global__ void __launch_bounds__(256, MINW) k_mul(uint32_t *out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (MODE == 0) {  // saturated CIOS (inlined)
        Fp<P> x, y;
        for (int i = 0; i < 8; ++i) { x.l[i] = seed * (i + 1) + tid; y.l[i] = seed + i * tid; }
        x.l[7] &= 0x0fffffff; y.l[7] &= 0x0fffffff;
        for (int it = 0; it < ITERS; ++it) { x = fp_mul(x, y); y = fp_mul(y, x); }
        uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= x.l[i] ^ y.l[i];
        out[tid] = s;
    } else {
        FpU<P> x, y;
        for (int i = 0; i < 9; ++i) { x.l[i] = (seed * (i + 1) + tid) & 0x1fffffff; y.l[i] = (seed + i * tid) & 0x1fffffff; }
        x.l[8] &= 0xffff; y.l[8] &= 0xffff;
        for (int it = 0; it < ITERS; ++it) {
            if constexpr (MODE == 1) { x = fpu_mul(x, y); y = fpu_mul(y, x); }
            if constexpr (MODE == 2) { x = fpu_sqr(x); y = fpu_sqr(y); }
            if constexpr (MODE == 3) { x = fpu_mul2(x, y); y = fpu_mul2(y, x); }
            if constexpr (MODE == 5) { x = fpu_mul_asm(x, y); y = fpu_mul_asm(y, x); }
            if constexpr (MODE == 4) { FpU<P> t = fpu_mul(x, y); FpU<P> u = fpu_mul(y, y); x = t; y = u; }  // 2 independent muls
        }
        uint32_t s = 0; for (int i = 0; i < 9; ++i) s ^= x.l[i] ^ y.l[i];
        out[tid] = s;
    }
}

template <int MODE, int MINW>
int run(const char *name, uint32_t *d, int blocks_per_cu) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * blocks_per_cu;
    k_mul<MODE, MINW><<<blocks, 256>>>(d, 1); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k_mul<MODE, MINW><<<blocks, 256>>>(d, 2 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double muls = (double)blocks * 256 * ITERS * 2;
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)k_mul<MODE, MINW>));
    double cyc = ms * 1e-3 * 2.4e9 * p.multiProcessorCount * 4 / (muls / 64);
    printf("%-34s blocks/CU=%d vgpr=%3d  %7.3f ms  %7.2f Gmul/s  %7.1f cyc/mul/SIMD@2.4GHz\n", name, blocks_per_cu, fa.numRegs, ms, muls / ms * 1e-6, cyc);
    return 0;
}

int main() {
    uint32_t *d; CHECK(hipMalloc(&d, (size_t)256 * 16 * 256 * 4));
    for (int bpc : {1, 2, 4, 8}) {
        run<0, 1>("sat CIOS 8x32 (inlined)", d, bpc);
        run<1, 1>("unsat 9x29 mul", d, bpc);
        run<2, 1>("unsat 9x29 sqr", d, bpc);
        run<3, 1>("unsat 9x29 mul, split acc", d, bpc);
        run<4, 1>("unsat 9x29 2 indep muls", d, bpc);
        run<5, 1>("unsat 9x29 mul, asm single chain", d, bpc);
    }
    return 0;
}
