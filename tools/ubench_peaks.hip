// Sustained product rates of every coordinate field in scope (gfx950): what `int_roofline.frac_of_measured` of each bench row
// is measured against. One dependent chain per thread (x <- x*y, y <- y*x) of the SAME product routines the accumulation
// kernels call - fsmul (signed lazy limbs, prime fields: madd_s) and f2s_mul (Fp2: madd_ts) - plus the dedicated squarings,
// at 1 / 2 / 4 / 8 workgroups per CU. Prints one JSON object on the last line (-> profiles/peaks_r06.json).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_peaks tools/ubench_peaks.hip && tools/ubench_peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#define GMSM_INLINE_MUL 1
#include "../gnark-crypto_amd/csrc/gmsm_curveu.h"
using namespace gmsm;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 1024;

template <class P>
__device__ __forceinline__ FpU<P> seed_elem(uint32_t a, uint32_t b) {
    FpU<P> x;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) x.l[i] = (a * (i + 1) + b * (i + 3)) & ((1u << (P::UW - 1)) - 1);
    x.l[P::UL - 1] &= 0xff;
    return x;
}

// MODE 0: prime-field product, 1: prime-field square, 2: Fp2 product, 3: Fp2 square (signed limbs: what the accumulation
// loops call); 4..7: the same four on unsigned limbs (fpu_mul / fpu_sqr / lz_mul / lz_sqr: the fix-up and reduction kernels,
// and the fastest bare product this code base has - the yardstick of int_roofline.frac_of_measured)
template <class P, int MODE>
__global__ void __launch_bounds__(256) k_chain(uint32_t *out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0;
    if constexpr (MODE < 2 || MODE == 4 || MODE == 5) {
        FpU<P> x = seed_elem<P>(seed, tid), y = seed_elem<P>(tid, seed + 7);
#pragma nounroll
        for (int it = 0; it < ITERS; ++it) {
            if constexpr (MODE == 0) { x = fsmul<true>(x, y); y = fsmul<true>(y, x); }
            else if constexpr (MODE == 1) { x = fssqr<true>(x); y = fssqr<true>(y); }
            else if constexpr (MODE == 4) { x = fmul<true>(x, y); y = fmul<true>(y, x); }
            else { x = fsqr<true>(x); y = fsqr<true>(y); }
        }
#pragma unroll
        for (int i = 0; i < P::UL; ++i) s ^= x.l[i] ^ y.l[i];
    } else {
        Fp2U<P> x{seed_elem<P>(seed, tid), seed_elem<P>(seed + 1, tid)}, y{seed_elem<P>(tid, seed + 7), seed_elem<P>(tid, seed + 9)};
        FpU<P> h;
#pragma nounroll
        for (int it = 0; it < ITERS; ++it) {
            if constexpr (MODE == 2) { x = f2s_mul<true>(x, y); y = f2s_mul<true>(y, x); }
            else if constexpr (MODE == 3) { x = f2s_sqr<true>(x, h); y = f2s_sqr<true>(y, h); }
            else if constexpr (MODE == 6) { x = lz_mul<true>(x, y); y = lz_mul<true>(y, x); }
            else { x = lz_sqr<true>(x); y = lz_sqr<true>(y); }
        }
#pragma unroll
        for (int i = 0; i < P::UL; ++i) s ^= x.a0.l[i] ^ y.a0.l[i] ^ x.a1.l[i] ^ y.a1.l[i];
    }
    out[tid] = s;
}

static std::string g_json;

template <class P, int MODE>
void run(const char *name, uint32_t *d) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    double best = 0;
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = p.multiProcessorCount * bpc;
        k_chain<P, MODE><<<blocks, 256>>>(d, 1);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) k_chain<P, MODE><<<blocks, 256>>>(d, 2 + r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        const double prods = (double)blocks * 256 * ITERS * 2;
        hipFuncAttributes fa;
        CHECK(hipFuncGetAttributes(&fa, (const void *)k_chain<P, MODE>));
        const double rate = prods / (ms * 1e-3);
        printf("%-28s blocks/CU=%d vgpr=%3d scratch=%4zu  %8.3f ms  %8.2f G products/s\n", name, bpc, fa.numRegs, (size_t)fa.localSizeBytes, ms, rate * 1e-9);
        if (rate > best) best = rate;
    }
    char buf[160];
    snprintf(buf, sizeof buf, "%s\"%s\": %.4g", g_json.empty() ? "" : ", ", name, best);
    g_json += buf;
}

int main() {
    uint32_t *d;
    CHECK(hipMalloc(&d, (size_t)256 * 16 * 256 * 4));
    run<bn254_fp_params, 0>("bn254_fp_mul", d);
    run<bn254_fp_params, 1>("bn254_fp_sqr", d);
    run<bn254_fp_params, 2>("bn254_fp2_mul", d);
    run<bn254_fp_params, 3>("bn254_fp2_sqr", d);
    run<bls12_381_fp_params, 0>("bls12_381_fp_mul", d);
    run<bls12_381_fp_params, 1>("bls12_381_fp_sqr", d);
    run<bls12_381_fp_params, 2>("bls12_381_fp2_mul", d);
    run<bls12_381_fp_params, 3>("bls12_381_fp2_sqr", d);
    run<bw6_761_fp_params, 0>("bw6_761_fp_mul", d);
    run<bw6_761_fp_params, 1>("bw6_761_fp_sqr", d);
    run<bn254_fp_params, 4>("bn254_fp_mul_unsigned", d);
    run<bn254_fp_params, 5>("bn254_fp_sqr_unsigned", d);
    run<bn254_fp_params, 6>("bn254_fp2_mul_unsigned", d);
    run<bn254_fp_params, 7>("bn254_fp2_sqr_unsigned", d);
    run<bls12_381_fp_params, 4>("bls12_381_fp_mul_unsigned", d);
    run<bls12_381_fp_params, 5>("bls12_381_fp_sqr_unsigned", d);
    run<bls12_381_fp_params, 6>("bls12_381_fp2_mul_unsigned", d);
    run<bls12_381_fp_params, 7>("bls12_381_fp2_sqr_unsigned", d);
    run<bw6_761_fp_params, 4>("bw6_761_fp_mul_unsigned", d);
    run<bw6_761_fp_params, 5>("bw6_761_fp_sqr_unsigned", d);
    printf("{%s}\n", g_json.c_str());
    return 0;
}
