// Does the instruction cache bound the wide-field kernels? 28-limb lazy Montgomery products (BW6-761), one wave per SIMD,
// loop bodies of 1, 2, 5 and 10 inlined products (13 KB of code each): cycles per product should not depend on the
// body size unless the loop streams its code from L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../gnark-crypto_amd/csrc/gmsm_fieldu.h"
using namespace gmsm;
using P = bw6_761_fp_params;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int TOTAL = 640;  // products per thread

template <int BODY>
__global__ void __launch_bounds__(256, 1) k_mulw(uint32_t *out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    FpU<P> x, y;
    for (int i = 0; i < P::UL; ++i) { x.l[i] = (seed * (i + 1) + tid) & 0x0fffffffu; y.l[i] = (seed + i * tid) & 0x0fffffffu; }
    x.l[P::UL - 1] &= 0xff; y.l[P::UL - 1] &= 0xff;
#pragma nounroll
    for (int it = 0; it < TOTAL / BODY; ++it) {
#pragma unroll
        for (int j = 0; j < BODY; ++j) {
            if (j & 1) y = fpu_mul(y, x); else x = fpu_mul(x, y);
        }
    }
    uint32_t s = 0; for (int i = 0; i < P::UL; ++i) s ^= x.l[i] ^ y.l[i];
    out[tid] = s;
}

template <int BODY>
int run(uint32_t *d, int blocks_per_cu) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * blocks_per_cu;
    k_mulw<BODY><<<blocks, 256>>>(d, 1); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k_mulw<BODY><<<blocks, 256>>>(d, 2 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double muls = (double)blocks * 256 * TOTAL;
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)k_mulw<BODY>));
    double cyc = ms * 1e-3 * 2.4e9 * p.multiProcessorCount * 4 / (muls / 64);
    printf("28-limb product, %2d inlined per loop body  blocks/CU=%d vgpr=%3d  %7.3f ms  %7.2f Gmul/s  %8.1f cyc/mul/SIMD@2.4GHz (1596 multiplies each)\n",
           BODY, blocks_per_cu, fa.numRegs, ms, muls / ms * 1e-6, cyc);
    return 0;
}

int main() {
    uint32_t *d; CHECK(hipMalloc(&d, (size_t)256 * 8 * 256 * 4));
    for (int bpc : {1, 2}) { run<1>(d, bpc); run<2>(d, bpc); run<5>(d, bpc); run<10>(d, bpc); }
    return 0;
}
