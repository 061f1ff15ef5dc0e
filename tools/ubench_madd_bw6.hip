// Mixed-addition throughput micro-benchmark (gfx950): the madd_u code of the accumulation kernel with its operands in
// registers (no gather, no bucket bookkeeping), at 1..3 waves per SIMD. Gives the ALU-only cost of one mixed addition
// as compiled, to separate multiplier work from memory / boundary overhead in k_accumulate_seg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../gnark-crypto_amd/csrc/gmsm_curveu.h"
using namespace gmsm;
using P = bw6_761_fp_params;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 32;

template <int MINW, int MODE>
__global__ void __launch_bounds__(256, MINW) k_madd(uint32_t *out, const uint32_t *in, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZU<P> acc;
    FpU<P> px, py;
    for (int i = 0; i < P::UL; ++i) {
        const uint32_t m = i < P::UL - 1 ? 0x0fffffffu : 0xffu;
        acc.x.l[i] = (in[i % 64] + tid) & m; acc.y.l[i] = (in[(9 + i) % 64] ^ tid) & m;
        acc.zz.l[i] = (in[(18 + i) % 64] + seed) & m; acc.zzz.l[i] = (in[(27 + i) % 64] + tid * 3) & m;
        px.l[i] = (in[(36 + i) % 64] + tid) & m; py.l[i] = (in[(45 + i) % 64] + seed * tid) & m;
    }
    bool inf = false;
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE == 0) madd_u<P, true>(acc, inf, px, py, (it & 1) != 0);
        px.l[0] = (px.l[0] + acc.x.l[1]) & 0x0fffffffu;  // next "point" depends on the result: nothing hoistable
        py.l[1] = (py.l[1] ^ acc.y.l[2]) & 0x0fffffffu;
    }
    uint32_t s = inf;
    for (int i = 0; i < P::UL; ++i) s ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.zz.l[i] ^ acc.zzz.l[i];
    out[tid] = s;
}

template <int MINW, int MODE>
int run(const char *name, uint32_t *d, const uint32_t *in, int blocks_per_cu) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * blocks_per_cu;
    k_madd<MINW, MODE><<<blocks, 256>>>(d, in, 1); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k_madd<MINW, MODE><<<blocks, 256>>>(d, in, 2 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double madds = (double)blocks * 256 * ITERS;
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void *)k_madd<MINW, MODE>));
    double cyc = ms * 1e-3 * 2.4e9 * p.multiProcessorCount * 4 / (madds / 64);
    printf("%-28s blocks/CU=%d vgpr=%3d  %7.3f ms  %7.2f Gmadd/s  %8.1f cyc/wave-madd/SIMD@2.4GHz  -> 2^24 madds in %.3f ms\n", name,
           blocks_per_cu, fa.numRegs, ms, madds / ms * 1e-6, cyc, 16777216.0 / (madds / ms));
    return 0;
}

int main() {
    uint32_t *d, *in; CHECK(hipMalloc(&d, (size_t)256 * 16 * 256 * 4)); CHECK(hipMalloc(&in, 64 * 4));
    uint32_t h[64]; for (int i = 0; i < 64; ++i) h[i] = 0x12345678u * (i + 1);
    CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    for (int bpc : {1, 2}) {
        run<1, 0>("bw6 madd_u (1 wave/SIMD build)", d, in, bpc);
        
    }
    return 0;
}
