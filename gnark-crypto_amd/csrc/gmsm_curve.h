// Short-Weierstrass (a = 0) group law in extended-Jacobian (XYZZ) coordinates over a coordinate field F
// (Fp for G1 / BW6-761 G2, Fp2 for BN254 and BLS12-381 G2). Shared by the device kernels and the host fold.
//
// Replaces, with identical group semantics (every special case of the reference is kept: inf + P, P + P, P + (-P)):
//   g1JacExtended.addMixed / subMixed      ecc/bn254/g1.go:822-930   (EFD madd-2008-s)
//   doubleMixed / doubleNegMixed           ecc/bn254/g1.go:933-985
//   g1JacExtended.add                      ecc/bn254/g1.go:736-788   (EFD add-2008-s)
//   g1JacExtended.double                   ecc/bn254/g1.go:795-817   (EFD dbl-2008-s-1, a = 0)
//   SetInfinity (1,1,0,0) / IsInfinity     ecc/bn254/g1.go:688-699
//   unsafeFromJacExtended / FromJacobian   ecc/bn254/g1.go:726-731, :150-166
// G2 twins: ecc/bn254/g2.go:663-970.
#pragma once
#include "gmsm_field.h"

// Full additions / doublings are used by the bucket reduction and the host fold, never by the accumulation loop:
// they are real calls on the device (one copy per kernel instead of one per call site).
#if defined(__HIPCC__)
#define GMSM_GROUP_HD __host__ __device__ __noinline__
#else
#define GMSM_GROUP_HD inline
#endif

namespace gmsm {

template <class F>
struct Affine {
    F x, y;
    GMSM_HD bool is_infinity() const { return x.is_zero() && y.is_zero(); }  // (0,0), g1.go:178
};

template <class F>
struct Jac {
    F x, y, z;
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    GMSM_HD static XYZZ infinity() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
    GMSM_HD bool is_infinity() const { return zz.is_zero(); }
};

// [2](+-a), a affine and not infinity
template <class F>
GMSM_HD XYZZ<F> xyzz_double_mixed(const Affine<F> &a, bool negate) {
    F U = fp_dbl(a.y);
    if (negate) U = fp_neg(U);
    F V = fp_sqr(U);
    F W = fp_mul(U, V);
    F S = fp_mul(a.x, V);
    F XX = fp_sqr(a.x);
    F M = fp_add(fp_dbl(XX), XX);
    F S2 = fp_dbl(S);
    F L = fp_mul(W, a.y);
    XYZZ<F> r;
    r.x = fp_sub(fp_sqr(M), S2);
    r.y = fp_mul(fp_sub(S, r.x), M);
    r.y = negate ? fp_add(r.y, L) : fp_sub(r.y, L);
    r.zz = V;
    r.zzz = W;
    return r;
}

// p += (negate ? -a : a)
template <class F>
GMSM_HD void xyzz_add_mixed(XYZZ<F> &p, const Affine<F> &a, bool negate) {
    if (a.is_infinity()) return;
    const F ay = negate ? fp_neg(a.y) : a.y;
    if (p.zz.is_zero()) {
        p.x = a.x;
        p.y = ay;
        p.zz = F::one();
        p.zzz = F::one();
        return;
    }
    F P = fp_sub(fp_mul(a.x, p.zz), p.x);
    F R = fp_sub(fp_mul(ay, p.zzz), p.y);
    if (P.is_zero()) {
        if (R.is_zero()) {
            p = xyzz_double_mixed(a, negate);
        } else {
            p.zz = F::zero();
            p.zzz = F::zero();
        }
        return;
    }
    F PP = fp_sqr(P);
    F PPP = fp_mul(P, PP);
    F Q = fp_mul(p.x, PP);
    F X3 = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
    F Y3 = fp_mul(fp_sub(Q, X3), R);
    p.y = fp_sub(Y3, fp_mul(p.y, PPP));
    p.x = X3;
    p.zz = fp_mul(p.zz, PP);
    p.zzz = fp_mul(p.zzz, PPP);
}

template <class F>
GMSM_GROUP_HD XYZZ<F> xyzz_double(const XYZZ<F> &q) {
    F U = fp_dbl(q.y);
    F V = fp_sqr(U);
    F W = fp_mul(U, V);
    F S = fp_mul(q.x, V);
    F XX = fp_sqr(q.x);
    F M = fp_add(fp_dbl(XX), XX);
    U = fp_mul(W, q.y);
    XYZZ<F> r;
    r.x = fp_sub(fp_sub(fp_sqr(M), S), S);
    r.y = fp_sub(fp_mul(fp_sub(S, r.x), M), U);
    r.zz = fp_mul(V, q.zz);
    r.zzz = fp_mul(W, q.zzz);
    return r;
}

// p += q
template <class F>
GMSM_GROUP_HD void xyzz_add(XYZZ<F> &p, const XYZZ<F> &q) {
    if (q.zz.is_zero()) return;
    if (p.zz.is_zero()) {
        p = q;
        return;
    }
    F U2 = fp_mul(q.x, p.zz);
    F U1 = fp_mul(p.x, q.zz);
    F A = fp_sub(U2, U1);
    F S2 = fp_mul(q.y, p.zzz);
    F S1 = fp_mul(p.y, q.zzz);
    F B = fp_sub(S2, S1);
    if (A.is_zero()) {
        if (B.is_zero()) {
            p = xyzz_double(q);
        } else {
            p.zz = F::zero();
            p.zzz = F::zero();
        }
        return;
    }
    F PP = fp_sqr(A);
    F PPP = fp_mul(A, PP);
    F Q = fp_mul(U1, PP);
    F V = fp_mul(S1, PPP);
    F X3 = fp_sub(fp_sub(fp_sub(fp_sqr(B), PPP), Q), Q);
    p.y = fp_sub(fp_mul(fp_sub(Q, X3), B), V);
    p.x = X3;
    p.zz = fp_mul(fp_mul(p.zz, q.zz), PP);
    p.zzz = fp_mul(fp_mul(p.zzz, q.zzz), PPP);
}

// XYZZ -> Jacobian (X*ZZ^2, Y*ZZZ^2, ZZZ); infinity -> (1,1,0)
template <class F>
GMSM_HD Jac<F> jac_from_xyzz(const XYZZ<F> &q) {
    if (q.zz.is_zero()) return Jac<F>{F::one(), F::one(), F::zero()};
    Jac<F> r;
    r.x = fp_mul(fp_sqr(q.zz), q.x);
    r.y = fp_mul(fp_sqr(q.zzz), q.y);
    r.z = q.zzz;
    return r;
}

// 2 q in Jacobian coordinates, a = 0 (dbl-2009-l; the formulas of (*G1Jac).DoubleAssign, g1.go:410-448): 2M + 5S where
// the XYZZ doubling takes 6M + 3S - the long runs of doublings of the host-side window fold use it. (Issuing the
// formula's independent products side by side, rows interleaved, was measured on the host: 248 against 260 ns per
// doubling - the 64-bit CIOS is bound by multiplier throughput, not by its carry chain.) q at infinity
// (z = 0) stays at infinity (z3 = 2 y z = 0), and so does a point with y = 0.
template <class F>
GMSM_HD Jac<F> jac_double(const Jac<F> &q) {
    const F A = fp_sqr(q.x), B = fp_sqr(q.y), C = fp_sqr(B);
    const F D = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(q.x, B)), A), C));
    const F E = fp_add(fp_dbl(A), A), Fq = fp_sqr(E);
    Jac<F> r;
    r.z = fp_dbl(fp_mul(q.y, q.z));
    r.x = fp_sub(Fq, fp_dbl(D));
    const F C8 = fp_dbl(fp_dbl(fp_dbl(C)));
    r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), C8);
    return r;
}

// Jacobian -> XYZZ (ZZ = Z^2, ZZZ = Z^3; X, Y unchanged)
template <class F>
GMSM_HD XYZZ<F> xyzz_from_jac(const Jac<F> &q) {
    XYZZ<F> r;
    r.x = q.x;
    r.y = q.y;
    r.zz = fp_sqr(q.z);
    r.zzz = fp_mul(r.zz, q.z);
    return r;
}

// Jacobian -> affine, infinity -> (0,0)
template <class F>
GMSM_HD Affine<F> affine_from_jac(const Jac<F> &q) {
    if (q.z.is_zero()) return Affine<F>{F::zero(), F::zero()};
    F a = fp_inv(q.z);
    F b = fp_sqr(a);
    Affine<F> r;
    r.x = fp_mul(q.x, b);
    r.y = fp_mul(fp_mul(q.y, b), a);
    return r;
}

}  // namespace gmsm
