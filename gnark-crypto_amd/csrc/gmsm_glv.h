// GLV half-scalars on the device: every scalar s is split into (k1, k2) with k1 + k2 lambda = s mod r and |k1|, |k2| <
// 2^GLV_BITS (about half of the scalar field's bits), and s P becomes k1 P + k2 phi(P), phi(x, y) = (w x, y) = [lambda](x, y).
// The MultiExp of n points then is one of 2 n entries (P_i, phi(P_i)) with half as many windows: the same additions in the
// buckets, half the bucket sets to reduce and half the doublings of the host's Horner fold - which is a serial chain of
// (nwin - 1) c doublings after the device has finished, 44 % of a small call.
//
// Replaces (as the reference uses it for scalar multiplication, never for MultiExp): ecc.SplitScalar / PrecomputeLattice,
// ecc/utils.go:62-170; (*G1Jac).mulGLV and phi, ecc/bn254/g1.go:529-600; lambdaGLV / thirdRootOneG1 / glvBasis,
// ecc/bn254/bn254.go:131-135. The constants are derived in gnark-crypto_amd/curves.py (GlvParams: the reference's lattice
// walk, our own rounding precision) and checked there; ANY rounding gives the congruence, so the result - the group element
// sum s_i P_i - is the one the reference computes, and the affine limbs the host hands out are bit-identical.
#pragma once
#include "gmsm_kernels.h"

namespace gmsm {

// out = words SHIFT .. SHIFT + NO - 1 of a[NA] * B[NBL] (schoolbook rows, carries from the low words included)
template <int NA, int NBL, int SHIFT, int NO, class GetB>
__device__ __forceinline__ void glv_mul_shift(const uint32_t (&a)[NA], GetB b, uint32_t (&out)[NO]) {
    uint32_t t[NA + NBL];
#pragma unroll
    for (int k = 0; k < NA + NBL; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            const uint64_t v = (uint64_t)a[i] * b(j) + t[i + j] + carry;
            t[i + j] = (uint32_t)v;
            carry = (uint32_t)(v >> 32);
        }
        t[i + NBL] = carry;
    }
#pragma unroll
    for (int k = 0; k < NO; ++k) out[k] = (SHIFT + k < NA + NBL) ? t[SHIFT + k < NA + NBL ? SHIFT + k : 0] : 0u;
}

// acc -= m * A mod 2^(32 HL)
template <int HL, class GetA>
__device__ __forceinline__ void glv_sub_lowmul(uint32_t (&acc)[HL], const uint32_t (&m)[HL], GetA a) {
    uint32_t p[HL];
#pragma unroll
    for (int k = 0; k < HL; ++k) p[k] = 0;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < HL - i; ++j) {
            const uint64_t v = (uint64_t)m[i] * a(j) + p[i + j] + carry;
            p[i + j] = (uint32_t)v;
            carry = (uint32_t)(v >> 32);
        }
    }
    uint32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < HL; ++k) acc[k] = __builtin_subc(acc[k], p[k], borrow, &borrow);
}

template <int HL>
__device__ __forceinline__ bool glv_abs(uint32_t (&k)[HL]) {  // two's complement -> magnitude; returns the sign
    const bool neg = (k[HL - 1] >> 31) != 0;
    uint32_t carry = neg ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
        const uint32_t v = neg ? ~k[i] : k[i];
        k[i] = v + carry;
        carry = (carry && k[i] == 0u) ? 1u : 0u;
    }
    return neg;
}

// s (regular form, < r) -> magnitudes and signs of k1, k2
template <class FrP>
__device__ __forceinline__ void glv_split(const uint32_t (&s)[FrP::N], uint32_t (&k1)[FrP::GLV_HL], bool &neg1,
                                          uint32_t (&k2)[FrP::GLV_HL], bool &neg2) {
    constexpr int HL = FrP::GLV_HL;
    uint32_t m1[HL], m2[HL];
    glv_mul_shift<FrP::N, FrP::GLV_NB, FrP::GLV_SH, HL>(s, [](int j) { return FrP::GLV_B1[j]; }, m1);
    glv_mul_shift<FrP::N, FrP::GLV_NB, FrP::GLV_SH, HL>(s, [](int j) { return FrP::GLV_B2[j]; }, m2);
#pragma unroll
    for (int k = 0; k < HL; ++k) {
        k1[k] = s[k];
        k2[k] = 0;
    }
    glv_sub_lowmul<HL>(k1, m1, [](int j) { return FrP::GLV_A11[j]; });
    glv_sub_lowmul<HL>(k1, m2, [](int j) { return FrP::GLV_A21[j]; });
    glv_sub_lowmul<HL>(k2, m1, [](int j) { return FrP::GLV_A12[j]; });
    glv_sub_lowmul<HL>(k2, m2, [](int j) { return FrP::GLV_A22[j]; });
    neg1 = glv_abs<HL>(k1);
    neg2 = glv_abs<HL>(k2);
}

// the digit code of -d from the code of d (0 stays 0): d > 0 -> 2d, d < 0 -> 2(-d - 1) + 1
__device__ __forceinline__ uint32_t code_negate(uint32_t code) {
    if (code == 0u) return 0u;
    return (code & 1u) ? (((code >> 1) + 1u) << 1) : ((((code >> 1) - 1u) << 1) | 1u);
}

// Window geometry of the half scalars (same rules as computeNbChunks / lastC, multiexp.go:681-693, over GLV_BITS bits)
template <class FrP>
struct GlvWindows {
    static constexpr uint32_t nwin(uint32_t c) { return ((uint32_t)FrP::GLV_BITS + c - 1) / c; }
};

// x -> w x for the coordinate types (phi: the y coordinate stays)
template <bool INL, class P>
__device__ __forceinline__ FpU<P> glv_mul_w(const FpU<P> &x, const FpU<P> &w) { return fmul<INL>(x, w); }
template <bool INL, class P>
__device__ __forceinline__ Fp2U<P> glv_mul_w(const Fp2U<P> &x, const FpU<P> &w) { return Fp2U<P>{fmul<INL>(x.a0, w), fmul<INL>(x.a1, w)}; }
template <class U, class C, bool INL>
__device__ __forceinline__ FpU<typename U::Params> glv_w() {  // GLV_W of the group in the lazy domain
    using P = typename U::Params;
    Fp<P> w;
#pragma unroll
    for (int i = 0; i < P::N; ++i) w.l[i] = C::GLV_W[i];
    return fpu_from_sat<P, INL>(w);
}

// ------------------------------------------------------------------ the pipeline's front end with GLV
// k_convert_points for 2 n entries: upoints[2 i] = P_i, upoints[2 i + 1] = phi(P_i), both in the lazy Montgomery domain.
// (0, 0) stays (0, 0) in both: the accumulation recognises infinity from the record itself.
// The two entries of a point are NEIGHBOURS (one 128-byte line for BN254 G1), not n records apart: with the phi copies in a
// second half of the array, scalars that are all equal - every first-half entry in one bucket, every second-half entry in
// another, both walked in index order - made the two walks hit addresses exactly 2^26 bytes apart at the same time, and
// k_accumulate_seg took 2.7-5.5 ms instead of 1.2 at 2^20 (profiles/r06_glv_all_equal.log).
template <class U, class C>
__global__ void __launch_bounds__(256) k_convert_points_glv(const void *__restrict__ points, size_t n, void *__restrict__ upoints) {
    using T = LzTraits<U>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<typename T::Sat> a = load_struct<Affine<typename T::Sat>>(points, i);
    const U ux = T::template from_sat<true>(a.x), uy = T::template from_sat<true>(a.y);
    UAffine<U> u;
    T::pack(ux, u.x);
    T::pack(uy, u.y);
    store_struct(upoints, 2 * i, u);
    T::pack(glv_mul_w<true>(ux, glv_w<U, C, true>()), u.x);
    store_struct(upoints, 2 * i + 1, u);
}

// k_decompose over the half scalars: digits is [nwin_local][2 n], entry 2 i = (P_i, k1), entry 2 i + 1 = (phi(P_i), k2); the
// sign of a half goes into its digit codes. One thread per scalar, the window width a template parameter (as k_decompose_c).
template <class FrP, class D, int C>
__global__ void __launch_bounds__(256) k_decompose_glv(const uint32_t *__restrict__ scalars, size_t n, WindowPlan plan,
                                                       D *__restrict__ digits, const uint8_t *__restrict__ skip) {
    constexpr int NR = FrP::N, HL = FrP::GLV_HL;
    constexpr uint32_t NW = ((uint32_t)FrP::GLV_BITS + C - 1) / C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrP> s;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scalars + i * NR);
        uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
    }
    const bool zero = s.is_zero() || (skip != nullptr && skip[i] != 0);
    s = fp_from_mont(s);
    uint32_t k[2][HL];
    bool neg[2];
    glv_split<FrP>(s.l, k[0], neg[0], k[1], neg[1]);
    constexpr uint32_t mask = (1u << C) - 1u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // digits of |k| in [-2^(C-1), 2^(C-1) - 1] as for a full scalar (multiexp.go:779-786) - or, when the half is negative and
        // every digit is about to change its sign, in [-(2^(C-1) - 1), 2^(C-1)]: the negated digits then lie in the usual range
        // and their codes fit the same 16 bits (a digit +2^(C-1) would have the code 2^C)
        const int max = (1 << (C - 1)) - 1 + (neg[h] ? 1 : 0);
        int carry = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w) {
            const uint32_t bit = w * C, idx = bit >> 5, sh = bit & 31;
            const uint64_t lo = idx < (uint32_t)HL ? k[h][idx < (uint32_t)HL ? idx : 0] : 0u;
            const uint64_t hi = idx + 1 < (uint32_t)HL ? k[h][idx + 1 < (uint32_t)HL ? idx + 1 : 0] : 0u;
            const uint64_t v = ((hi << 32) | lo) >> sh;
            int digit = carry + (int)((uint32_t)v & mask);
            uint32_t code;
            if (w + 1 < NW) {
                carry = 0;
                if (digit > max) {
                    digit -= 1 << C;
                    carry = 1;
                }
                code = digit == 0 ? 0u : (digit > 0 ? ((uint32_t)digit << 1) : ((((uint32_t)(-digit) - 1u) << 1) | 1u));
            } else {
                code = (uint32_t)digit << 1;  // top window: no borrow
            }
            if (neg[h]) code = code_negate(code);
            if (w >= plan.win_first && (w - plan.win_first) % plan.win_stride == 0) {
                const uint32_t kk = (w - plan.win_first) / plan.win_stride;
                if (kk < plan.nwin_local) digits[(size_t)kk * (2 * n) + 2 * i + (size_t)h] = (D)(zero ? 0u : code);
            }
        }
    }
}

// test hook: out[i] = {neg1, k1[HL], neg2, k2[HL]} (uint32 words) for scalar i (Montgomery form, as everywhere)
template <class FrP>
__global__ void __launch_bounds__(256) k_glv_split_debug(const uint32_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ out) {
    constexpr int NR = FrP::N, HL = FrP::GLV_HL;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrP> s;
#pragma unroll
    for (int k = 0; k < NR; ++k) s.l[k] = scalars[i * NR + k];
    s = fp_from_mont(s);
    uint32_t k1[HL], k2[HL];
    bool n1, n2;
    glv_split<FrP>(s.l, k1, n1, k2, n2);
    uint32_t *o = out + i * (2 * (HL + 1));
    o[0] = n1 ? 1u : 0u;
    o[HL + 1] = n2 ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < HL; ++k) {
        o[1 + k] = k1[k];
        o[HL + 2 + k] = k2[k];
    }
}

}  // namespace gmsm
