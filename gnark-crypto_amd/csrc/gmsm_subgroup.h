// IsInSubGroup the way the reference decides it: one endomorphism identity per group instead of the 255/377-bit [r]P walk.
//
//   BN254 G1      prime order: on the curve is in the group                         ecc/bn254/g1.go:475-482
//   BN254 G2      a = [x]P:  2 psi^3(a) - (psi^2(a) + psi(a) + a + P) = 0            ecc/bn254/g2.go:483-497
//   BLS12-381 G1  [x]([x] phi(P)) + P = 0                                           ecc/bls12-381/g1.go:481-492
//   BLS12-381 G2  [x]P + psi(P) = 0                                                 ecc/bls12-381/g2.go:484-491
//   BW6-761 G1/G2 [x]P + P + [x^2]([x] phi(P) - phi(P)) + phi(P) = 0                ecc/bw6-761/g1.go:482-496, g2.go:488-502
// with phi(x, y) = (w x, y) (w^3 = 1; g1.go:530-534), psi(x, y) = (conj(x) u, conj(y) v) (g2.go:528-534) and x = xGen, a
// 63/64-bit constant of low weight: 63-252 doublings and a handful of additions against 254-376 doublings and ~half as many
// additions of the definition. The predicate is the same on every point of the curve (what the reference relies on;
// tests/test_subgroup_model.py checks the identities against [r]P on torsion points of the cofactor in big integers, the GPU
// suite checks this file against point_in_r_torsion on the device) - the definition stays reachable as check level 3.
//
// One lane per point on the lazy limbs of the MSM pipeline (gmsm_curveu.h): mixed additions while the base is affine,
// add-2008-s once it is a computed multiple. The reference's mulWindowed is a 2-bit window; a constant of weight 6-7 does
// not need one, and BN254's x (weight 28) pays 28 mixed additions either way. Special cases as in point_in_r_torsion: a
// doubling that lands on infinity (2-torsion) is caught by the exact zero test of zz, P + (-P) inside the additions.
#pragma once
#include "gmsm_curveu.h"

namespace gmsm {

template <class F> struct IngestLazy;
template <class P> struct IngestLazy<Fp<P>> { using type = FpU<P>; };
template <class P> struct IngestLazy<Fp2<P>> { using type = Fp2U<P>; };
template <class P> __device__ __forceinline__ bool ingest_zz_is_zero(const FpU<P> &zz) { return fpu_prod_is_zero(zz); }   // a product: < 3q
template <class P> __device__ __forceinline__ bool ingest_zz_is_zero(const Fp2U<P> &zz) { return lz_is_zero(zz); }       // class R

template <class U>
struct SgPoint {  // a computed multiple: extended-Jacobian coordinates + the infinity flag
    XYZZL<U> v;
    bool inf;
};

template <bool INL, class U>
__device__ __forceinline__ void sg_dbl(SgPoint<U> &a) {
    if (a.inf) return;
    a.v = lz_pdbl<INL>(a.v);
    if (ingest_zz_is_zero(a.v.zz)) a.inf = true;
}
// a += (+-)(bx, by), affine
template <bool INL, class U>
__device__ __forceinline__ void sg_madd(SgPoint<U> &a, const U &bx, const U &by, bool negate) {
    lz_madd<INL>(a.v, a.inf, bx, by, negate);
    if (!a.inf && ingest_zz_is_zero(a.v.zz)) a.inf = true;  // the doubling branch on a 2-torsion point
}
template <bool INL, class U>
__device__ __forceinline__ void sg_add(SgPoint<U> &a, const SgPoint<U> &b) {
    lz_padd<INL>(a.v, a.inf, b.v, b.inf);
    if (!a.inf && ingest_zz_is_zero(a.v.zz)) a.inf = true;
}

// [x](bx, by), x a 64-bit constant, most significant bit first
template <bool INL, class U>
__device__ SgPoint<U> sg_mul_x_affine(const U &bx, const U &by, unsigned long long x) {
    SgPoint<U> acc;
    acc.inf = true;
#pragma nounroll
    for (int bit = 63; bit >= 0; --bit) {
        sg_dbl<INL>(acc);
        if ((x >> bit) & 1ull) sg_madd<INL>(acc, bx, by, false);
    }
    return acc;
}
// [x]b for a computed b
template <bool INL, class U>
__device__ SgPoint<U> sg_mul_x(const SgPoint<U> &b, unsigned long long x) {
    SgPoint<U> acc;
    acc.inf = true;
    if (b.inf) return acc;
#pragma nounroll
    for (int bit = 63; bit >= 0; --bit) {
        sg_dbl<INL>(acc);
        if ((x >> bit) & 1ull) sg_add<INL>(acc, b);
    }
    return acc;
}

template <class U, class C, bool INL>
__device__ __forceinline__ U sg_const_w() {  // thirdRootOne of the group, lazy domain
    using T = LzTraits<U>;
    typename T::Sat w;
#pragma unroll
    for (int i = 0; i < T::Params::N; ++i) w.l[i] = C::ENDO_W[i];
    return T::template from_sat<INL>(w);
}
template <class P, bool INL>
__device__ __forceinline__ Fp2U<P> sg_const_fp2(const uint32_t *words) {
    Fp2<P> s;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        s.a0.l[i] = words[i];
        s.a1.l[i] = words[P::N + i];
    }
    return LzTraits<Fp2U<P>>::template from_sat<INL>(s);
}
template <class P>
__device__ __forceinline__ Fp2U<P> sg_conj(const Fp2U<P> &a) {  // class R -> class R
    return Fp2U<P>{a.a0, fpu_subr(lz_zero((const FpU<P> *)nullptr), a.a1)};
}
// psi on extended-Jacobian coordinates: x = X/ZZ, y = Y/ZZZ, so (conj(X) u, conj(Y) v, conj(ZZ), conj(ZZZ))
template <bool INL, class P>
__device__ __forceinline__ void sg_psi(SgPoint<Fp2U<P>> &a, const Fp2U<P> &u, const Fp2U<P> &v) {
    if (a.inf) return;
    a.v.x = lz_mul<INL>(sg_conj(a.v.x), u);
    a.v.y = lz_mul<INL>(sg_conj(a.v.y), v);
    a.v.zz = sg_conj(a.v.zz);
    a.v.zzz = sg_conj(a.v.zzz);
}

// The reference's IsInSubGroup for a point that IS on the curve and is not infinity.
template <class F, class C>
__device__ __forceinline__ bool point_in_subgroup_endo(const Affine<F> &a) {  // inlined into the kernel: as a 142 KB FUNCTION it met the long-branch trap (tools/check_long_branch.py)
    using U = typename IngestLazy<F>::type;
    using T = LzTraits<U>;
    // Every group inlines its group operations here (round 6, same-box A/B profiles/r06_subgroup_inline_ab.log: BN254 G2 2^20 41.4 ->
    // 22.1 ms, BLS12-381 G2 2^18 12.9 -> 9.7, BW6-761 2^18 57.0 -> 47.2 against the out-of-line operations the level-3 walk and the
    // fixed-base kernels keep for code size): a few hundred KB of straight-line code per kernel, as in the accumulation.
    // -DGMSM_SUBGROUP_INL_BYTES=56 gives the out-of-line form back for the types wider than 14 words (tools/build_ab.sh).
#ifndef GMSM_SUBGROUP_INL_BYTES
#define GMSM_SUBGROUP_INL_BYTES (28 * 4)
#endif
    constexpr bool INL = sizeof(U) <= GMSM_SUBGROUP_INL_BYTES;
    constexpr unsigned long long X = C::X_GEN;
    static_assert(C::SUBGROUP_TEST >= 1 && C::SUBGROUP_TEST <= 4, "no endomorphism test for this group");
    const U px = T::template from_sat<INL>(a.x), py = T::template from_sat<INL>(a.y);
    if constexpr (C::SUBGROUP_TEST == 1) {  // BLS12-381 G1
        const U phx = lz_mul<INL>(px, sg_const_w<U, C, INL>());
        SgPoint<U> r = sg_mul_x_affine<INL>(phx, py, X);
        r = sg_mul_x<INL>(r, X);
        sg_madd<INL>(r, px, py, false);
        return r.inf;
    } else if constexpr (C::SUBGROUP_TEST == 2) {  // BLS12-381 G2
        using P = typename U::Params;
        SgPoint<U> r = sg_mul_x_affine<INL>(px, py, X);
        const U sx = lz_mul<INL>(sg_conj(px), sg_const_fp2<P, INL>(C::ENDO_U));
        const U sy = lz_mul<INL>(sg_conj(py), sg_const_fp2<P, INL>(C::ENDO_V));
        sg_madd<INL>(r, sx, sy, false);
        return r.inf;
    } else if constexpr (C::SUBGROUP_TEST == 3) {  // BN254 G2
        using P = typename U::Params;
        const U eu = sg_const_fp2<P, INL>(C::ENDO_U), ev = sg_const_fp2<P, INL>(C::ENDO_V);
        SgPoint<U> A = sg_mul_x_affine<INL>(px, py, X);  // a = [x]P
        SgPoint<U> r = A;
        sg_psi<INL>(r, eu, ev);                          // b = psi(a)
        sg_madd<INL>(A, px, py, false);                  // a += P
        SgPoint<U> c = r;                                // c = b ...
        sg_psi<INL>(r, eu, ev);                          // res = psi(b)
        sg_add<INL>(c, r);                               // ... + psi^2(a)
        sg_add<INL>(c, A);                               // ... + a + P
        sg_psi<INL>(r, eu, ev);                          // res = psi^3(a)
        sg_dbl<INL>(r);
        if (!c.inf) c.v.y = lz_sub(lz_zero((const U *)nullptr), c.v.y);
        sg_add<INL>(r, c);                               // 2 psi^3(a) - c
        return r.inf;
    } else {  // BW6-761, both groups
        const U phx = lz_mul<INL>(px, sg_const_w<U, C, INL>());
        SgPoint<U> r = sg_mul_x_affine<INL>(phx, py, X);
        sg_madd<INL>(r, phx, py, true);                  // [x] phi(P) - phi(P)
        r = sg_mul_x<INL>(r, X);
        r = sg_mul_x<INL>(r, X);
        sg_madd<INL>(r, phx, py, false);
        SgPoint<U> t = sg_mul_x_affine<INL>(px, py, X);
        sg_madd<INL>(t, px, py, false);
        sg_add<INL>(t, r);
        return t.inf;
    }
}

}  // namespace gmsm
