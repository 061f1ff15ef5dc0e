// Fixed-base batch scalar multiplication and batch normalisation.
//
// Replaces: BatchScalarMultiplicationG1/G2 (ecc/bn254/g1.go:1039-1118, g2.go) and BatchJacobianToAffineG1
// (g1.go:989-1035). The reference walks every scalar with a windowed double-and-add over one table of 2^(c-1)
// multiples of the base (c doublings per window). Here the table holds the multiples of 2^(c*j) * base for EVERY
// window j (nwin * 2^(c-1) affine points, a few hundred KB: L2-resident), so a scalar costs nwin mixed additions and
// no doublings: 32 instead of 254 + 32 group operations for BN254 at c = 8. One thread per scalar; the results are
// normalised with the Montgomery trick, K points per thread and one Fermat inversion per K.
//
// Same digit code as the MultiExp path (k_decompose): 0 skip, d>0 -> 2d, d<0 -> 2(-d-1)+1; table[j][code_bucket].
#pragma once
#include "gmsm_kernels.h"

namespace gmsm {

// x^(q-2) on lazy limbs (operands < 13q are fine; the result is < 2q). q - 2 is read from the saturated modulus; q is
// odd and its lowest word is > 1 for every field in scope, so the subtraction never borrows.
template <class P>
__device__ __noinline__ FpU<P> fpu_inv(const FpU<P> &x) {
    static_assert(P::Q[0] > 1u, "q - 2 must not borrow out of the lowest word");
    FpU<P> r = x;
    bool started = false;
#pragma unroll
    for (int li = P::N - 1; li >= 0; --li) {
        const uint32_t w = li == 0 ? P::Q[li] - 2u : P::Q[li];
#pragma nounroll
        for (int b = 31; b >= 0; --b) {
            const bool bit = (w >> b) & 1u;
            if (started) {
                r = fsqr<false>(r);
                if (bit) r = fmul<false>(r, x);
            } else if (bit) {
                started = true;  // r = x
            }
        }
    }
    return r;
}

template <class P>
__device__ __forceinline__ FpU<P> lz_inv(const FpU<P> &x) {
    return fpu_inv(x);
}
// 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + a1^2)   (E2.Inverse, internal/fptower/e2_bn254.go:61-72)
template <class P>
__device__ __forceinline__ Fp2U<P> lz_inv(const Fp2U<P> &x) {
    const FpU<P> t = fpu_inv(fpu_addr(fsqr<false>(x.a0), fsqr<false>(x.a1)));
    Fp2U<P> r;
    r.a0 = fmul<false>(x.a0, t);
    r.a1 = fpu_subr(lz_zero((const FpU<P> *)nullptr), fmul<false>(x.a1, t));
    return r;
}

// One thread per scalar: rec[i] = sum_j digit_j(s_i) * table[j][.]  (lazy XYZZ record, infinity = zero zz limbs).
template <class U, class FrP, bool INL>
__global__ void __launch_bounds__(256) k_fixed_base(const uint32_t *__restrict__ scalars, size_t n, WindowPlan plan,
                                                    const void *__restrict__ table, void *__restrict__ recs) {
    using T = LzTraits<U>;
    constexpr int NR = FrP::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrP> s;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scalars + i * NR);
        uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
    }
    s = fp_from_mont(s);
    const uint32_t c = plan.c;
    const uint32_t mask = (1u << c) - 1u;
    const int max = (1 << (c - 1)) - 1;
    int carry = 0;
    XYZZL<U> acc;
    bool inf = true;
#pragma nounroll
    for (uint32_t w = 0; w < plan.nwin_total; ++w) {
        const uint32_t bit = w * c, idx = bit >> 5, sh = bit & 31;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {  // register-array select without dynamic indexing
            lo = (uint32_t)k == idx ? s.l[k] : lo;
            hi = (uint32_t)k == idx + 1 ? s.l[k] : hi;
        }
        const uint64_t v = (((uint64_t)hi << 32) | lo) >> sh;
        int digit = carry + (int)((uint32_t)v & mask);
        uint32_t code;
        if (w + 1 < plan.nwin_total) {
            carry = 0;
            if (digit > max) {
                digit -= 1 << c;
                carry = 1;
            }
            code = digit == 0 ? 0u : (digit > 0 ? ((uint32_t)digit << 1) : ((((uint32_t)(-digit) - 1u) << 1) | 1u));
        } else {
            code = (uint32_t)digit << 1;  // top window: no borrow (multiexp.go:788-800)
        }
        if (code) {
            const UAffine<U> p = load_struct<UAffine<U>>(table, (size_t)w * plan.nbuckets + code_bucket(code));
            lz_madd_acc<INL>(acc, inf, T::unpack(p.x), T::unpack(p.y), (code & 1u) != 0);
        }
    }
    lz_acc_finish(acc, inf);
    lazy_store<U>(recs, i, acc, inf);
}

// Jacobian (X, Y, Z) in Go layout -> the same lazy XYZZ record (ZZ = Z^2, ZZZ = Z^3), so that one normalisation kernel
// serves both entries.
template <class U, bool INL>
__global__ void __launch_bounds__(256) k_jac_to_recs(const void *__restrict__ jac, size_t n, void *__restrict__ recs) {
    using T = LzTraits<U>;
    using S = typename T::Sat;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Jac<S> p = load_struct<Jac<S>>(jac, i);
    XYZZL<U> r;
    const bool inf = p.z.is_zero();
    if (!inf) {
        const U z = T::template from_sat<INL>(p.z);
        r.x = T::template from_sat<INL>(p.x);
        r.y = T::template from_sat<INL>(p.y);
        r.zz = lz_sqr<INL>(z);
        r.zzz = lz_mul<INL>(r.zz, z);
    }
    lazy_store<U>(recs, i, r, inf);
}

// Montgomery's trick, K records per thread: prefix products of the ZZZ coordinates (spilled to `prefix`, n elements),
// one inversion, then back to front  u = 1/ZZZ_i,  x = X * ZZ^2 * u^2 (= X/ZZ because ZZ^3 = ZZZ^2),  y = Y * u.
// Output: Go-layout affine points (canonical Montgomery limbs), infinity = (0, 0)  (g1.go:1003-1006, 1013-1016).
template <class U, bool INL, int K>
__global__ void __launch_bounds__(64) k_batch_normalize(const void *__restrict__ recs, size_t n, U *__restrict__ prefix,
                                                        void *__restrict__ out_affine) {
    using T = LzTraits<U>;
    using S = typename T::Sat;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = g * K;
    if (i0 >= n) return;
    const size_t i1 = i0 + K < n ? i0 + K : n;
    U run = lz_one((const U *)nullptr);
#pragma nounroll
    for (size_t i = i0; i < i1; ++i) {
        const UnsatElem<U> e = unsat_load<U>(recs, i);
        prefix[i] = run;
        if (!e.inf) run = lz_mul<INL>(run, e.v.zzz);
    }
    U inv = lz_inv(run);
#pragma nounroll
    for (size_t i = i1; i-- > i0;) {
        const UnsatElem<U> e = unsat_load<U>(recs, i);
        Affine<S> a;
        a.x = S::zero();
        a.y = S::zero();
        if (!e.inf) {
            const U u = lz_mul<INL>(inv, prefix[i]);
            inv = lz_mul<INL>(inv, e.v.zzz);
            const U u2 = lz_sqr<INL>(u), zz2 = lz_sqr<INL>(e.v.zz);
            a.x = T::template to_sat<INL>(lz_mul<INL>(lz_mul<INL>(e.v.x, zz2), u2));
            a.y = T::template to_sat<INL>(lz_mul<INL>(e.v.y, u));
        }
        store_struct(out_affine, i, a);
    }
}

}  // namespace gmsm
