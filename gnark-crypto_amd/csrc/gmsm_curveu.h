// Extended-Jacobian group law on the unsaturated ("lazy") field representation of gmsm_fieldu.h, plus the arithmetic
// policy UnsatOps<U> the one-lane bucket kernels (k_fixup_seg, k_reduce_serial) are written against: Elem, load/store of
// the lazy XYZZ record in memory, store_final, add, dbl - once for every element type U (FpU<P> or Fp2U<P>). The kernels
// that combine few elements use the lane-quad forms of gmsm_quad.h instead.
//
// Same group semantics as the reference's g1JacExtended (ecc/bn254/g1.go:682-985): every special case is kept
// (inf + P, P + inf, P + P -> doubling, P + (-P) -> infinity). Value bounds are stated in multiples of q and hold for
// every field in scope because 2^(UL*UW)/q >= 169 (BN254 261-254 = 7 spare bits; BLS12-381 392-381 = 11; BW6-761 23):
//   stored coordinates: x < 11, y < 7, zz, zzz < 3;   mul(a,b) < a*b/169 + 1.
#pragma once
#include "gmsm_curve.h"
#include "gmsm_field2u.h"

namespace gmsm {

// Y3 of the additions is one double product with a single Montgomery reduction (fpu_mul_add): measured on BN254 G1 against
// the two-product form, k_accumulate_seg 1.385 -> 1.286 ms at 2^20 (-7 %), 20.96 -> 20.10 ms at 2^24 (round 2).

template <class U>
struct XYZZL {  // extended-Jacobian point over a lazy element type (FpU<P> or Fp2U<P>)
    U x, y, zz, zzz;
};
template <class P>
using XYZZU = XYZZL<FpU<P>>;

template <class P>
GMSM_HD FpU<P> fpu_one() {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = P::UONE[i];
    return r;
}

// cheap exact test "a == 0 mod q" for a normalised product-class value a < 3q
template <class P>
GMSM_HD bool fpu_prod_is_zero(const FpU<P> &a) {
    const uint32_t l0 = a.l[0];
    if (l0 == 0u || l0 == P::UQ1[0] || l0 == P::UQ2[0]) return fpu_is_zero_lt3q(a);
    return false;
}

// acc = [2](px, py), affine input (doubleMixed / doubleNegMixed, g1.go:933-985); px < 2, py < 6.
template <class P, bool INL = true>
GMSM_HD void double_mixed_u(XYZZU<P> &acc, const FpU<P> &px, const FpU<P> &py) {
    const FpU<P> U = fpu_dbl(py);                                       // < 12
    const FpU<P> V = fsqr<INL>(U);                                        // < 2
    const FpU<P> W = fmul<INL>(U, V);                                     // < 2
    const FpU<P> S = fmul<INL>(px, V);                                    // < 2
    const FpU<P> XX = fsqr<INL>(px);                                      // < 2
    const FpU<P> M = fpu_add(fpu_add(XX, XX), XX);                      // < 6
    const FpU<P> X3 = fpu_sub<P, 4>(fsqr<INL>(M), fpu_dbl(S));            // < 6
    const FpU<P> Y3 = fpu_sub<P, 4>(fmul<INL>(fpu_sub<P, 16>(S, X3), M), fmul<INL>(W, py));  // < 6
    acc.x = X3;
    acc.y = Y3;
    acc.zz = V;
    acc.zzz = W;
}

// acc += (+-)(px, py): addMixed / subMixed (g1.go:822-930, madd-2008-s). px, py_in < 2.
template <class P, bool INL = true>
GMSM_HD void madd_u(XYZZU<P> &acc, bool &inf, const FpU<P> &px, const FpU<P> &py_in, bool negate) {
    const FpU<P> py = negate ? fpu_neg4<P>(py_in) : py_in;           // < 6
    if (inf) {
        acc.x = px;
        acc.y = py;
        acc.zz = fpu_one<P>();
        acc.zzz = fpu_one<P>();
        inf = false;
        return;
    }
    const FpU<P> Pv = fpu_sub<P, 16>(fmul<INL>(px, acc.zz), acc.x);    // < 18
    const FpU<P> Rv = fpu_sub<P, 16>(fmul<INL>(py, acc.zzz), acc.y);   // < 18
    const FpU<P> PP = fsqr<INL>(Pv);                                   // < 3
    if (fpu_prod_is_zero(PP)) {                                      // same x (g1.go:846-854); Pv == 0 <=> Pv^2 == 0
        if (fpu_prod_is_zero(fsqr<INL>(Rv))) double_mixed_u<P, INL>(acc, px, py);  // P + P
        else inf = true;                                                    // P + (-P)
        return;
    }
    const FpU<P> PPP = fmul<INL>(Pv, PP);                              // < 2
    const FpU<P> Q = fmul<INL>(acc.x, PP);                             // < 2
    const FpU<P> RR = fsqr<INL>(Rv);                                   // < 3
    const FpU<P> X3 = fpu_sub_sub2<P>(RR, PPP, Q);                                              // < 3 + 8 = 11
    // Y3 = (Q - X3) R - y PPP as ONE reduced product: (Q - X3) R + (8q - y) PPP  (18*18 + 8*2 = 340 < 3*169 -> < 4)
    const FpU<P> Y3 = fmuladd<INL>(fpu_sub<P, 16>(Q, X3), Rv, fpu_neg8c<P>(acc.y), PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fmul<INL>(acc.zz, PP);
    acc.zzz = fmul<INL>(acc.zzz, PPP);
}

// The same mixed addition on SIGNED limbs (gmsm_fieldu.h, "signed limbs"): the form k_accumulate_seg and the fixed-base
// walk run. Differences are one subtraction per limb, and of madd_u's five carry passes one is left (X3); the products
// are the same product scans on signed columns. Values in multiples of q (A = 2^(L W)/q >= 169; a product of |a|, |b|
// lies in (-|a||b|/A, |a||b|/A + 1)):
//   px in [0, 2), py in (-2, 2);  accumulator: X in (-3.2, 6), Y in (-2, 6), ZZ in [0, 1.1), ZZZ in (-0.1, 1.1)
//   (the upper 6 only right after the doubling branch, which answers in madd_u's unsigned class)
//   P = px ZZ - X in (-6.1, 4.3)   R = py ZZZ - Y in (-6.1, 3.1)   PP in [0, 1.3)   RR in [0, 1.3)
//   PPP in (-0.1, 1.1)   Q = X PP in (-0.1, 1.1)   X3 = RR - PPP - 2Q in (-3.2, 1.5)
//   Y3 = ((Q - X3) R - Y PPP) / R' in (-0.2, 1.2)   [|Q - X3| < 4.4, one reduction for both products]
// Limbs: products come out normalised with a signed top limb, X3 is carry-passed, every difference of two such values
// has limbs within +-(2^W + 4): what the signed product scans admit. P == 0 mod q is tested on PP = P^2 >= 0 exactly as
// in madd_u. The accumulator leaves this class through lz_acc_finish (+4q on x, y, zzz: x < 11, y < 11, zz < 3, zzz < 6,
// non-negative, nearly normalised) - or, in k_accumulate_seg, is stored as it is, and the kernels that read the bucket
// and partial-sum records do the same on loading them (lz_rec_fresh).
template <class P, bool INL = true>
GMSM_HD void madd_s(XYZZU<P> &acc, bool &inf, const FpU<P> &px, const FpU<P> &py_in, bool negate) {
    const FpU<P> py = negate ? fps_neg<P>(py_in) : py_in;
    if (inf) {
        acc.x = px;
        acc.y = py;
        acc.zz = fpu_one<P>();
        acc.zzz = fpu_one<P>();
        inf = false;
        return;
    }
    const FpU<P> Pv = fps_sub<P>(fsmul<INL>(px, acc.zz), acc.x);
    const FpU<P> Rv = fps_sub<P>(fsmul<INL>(py, acc.zzz), acc.y);
    const FpU<P> PP = fssqr<INL>(Pv);
    if (fpu_prod_is_zero(PP)) {                                      // same x (g1.go:846-854)
        if (fpu_prod_is_zero(fssqr<INL>(Rv))) double_mixed_u<P, INL>(acc, px, negate ? fpu_neg4<P>(py_in) : py_in);  // P + P
        else inf = true;                                             // P + (-P)
        return;
    }
    const FpU<P> PPP = fsmul<INL>(Pv, PP);
    const FpU<P> Q = fsmul<INL>(acc.x, PP);
    const FpU<P> RR = fssqr<INL>(Rv);
    const FpU<P> X3 = fps_sub_sub2<P>(RR, PPP, Q);
    const FpU<P> Y3 = fsmuladd<INL>(fps_sub<P>(Q, X3), Rv, fps_neg<P>(acc.y), PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fsmul<INL>(acc.zz, PP);
    acc.zzz = fsmul<INL>(acc.zzz, PPP);
}

// r = [2]q (g1.go:795-817, dbl-2008-s-1, a = 0); q not infinity.
template <class P, bool INL = true>
GMSM_HD XYZZU<P> double_u(const XYZZU<P> &q) {
    const FpU<P> U = fpu_dbl(q.y);                                      // < 14
    const FpU<P> V = fsqr<INL>(U);                                        // < 3
    const FpU<P> W = fmul<INL>(U, V);                                     // < 2
    const FpU<P> S = fmul<INL>(q.x, V);                                   // < 2
    const FpU<P> XX = fsqr<INL>(q.x);                                     // < 2
    const FpU<P> M = fpu_add(fpu_add(XX, XX), XX);                      // < 6
    XYZZU<P> r;
    r.x = fpu_sub<P, 4>(fsqr<INL>(M), fpu_dbl(S));                        // < 6
    r.y = fpu_sub<P, 4>(fmul<INL>(fpu_sub<P, 16>(S, r.x), M), fmul<INL>(W, q.y));  // < 6
    r.zz = fmul<INL>(V, q.zz);
    r.zzz = fmul<INL>(W, q.zzz);
    return r;
}

// p += q (g1.go:736-788, add-2008-s), infinity carried as flags.
template <class P, bool INL = true>
GMSM_HD void add_u(XYZZU<P> &p, bool &pinf, const XYZZU<P> &q, bool qinf) {
    if (qinf) return;
    if (pinf) {
        p = q;
        pinf = false;
        return;
    }
    const FpU<P> U2 = fmul<INL>(q.x, p.zz);                               // < 2
    const FpU<P> U1 = fmul<INL>(p.x, q.zz);                               // < 2
    const FpU<P> S2 = fmul<INL>(q.y, p.zzz);                              // < 2
    const FpU<P> S1 = fmul<INL>(p.y, q.zzz);                              // < 2
    const FpU<P> A = fpu_sub<P, 4>(U2, U1);                             // < 6
    const FpU<P> B = fpu_sub<P, 4>(S2, S1);                             // < 6
    const FpU<P> PP = fsqr<INL>(A);                                       // < 2
    if (fpu_prod_is_zero(PP)) {
        if (fpu_prod_is_zero(fsqr<INL>(B))) p = double_u<P, INL>(q);
        else pinf = true;
        return;
    }
    const FpU<P> PPP = fmul<INL>(A, PP);                                  // < 2
    const FpU<P> Q = fmul<INL>(U1, PP);                                   // < 2
    const FpU<P> X3 = fpu_sub<P, 4>(fpu_sub<P, 4>(fsqr<INL>(B), PPP), fpu_dbl(Q));  // < 2 + 4 + 4
    // Y3 = (Q - X3) B - S1 PPP with one reduction: (Q - X3) B + (8q - S1) PPP  (18*6 + 8*2 = 124 -> < 2)
    p.y = fmuladd<INL>(fpu_sub<P, 16>(Q, X3), B, fpu_neg8c<P>(S1), PPP);
    p.x = X3;
    p.zz = fmul<INL>(fmul<INL>(p.zz, q.zz), PP);
    p.zzz = fmul<INL>(fmul<INL>(p.zzz, q.zzz), PPP);
}

// ------------------------------------------------------------------ generic group law on the reduced class
// Same formulas on the lz_ interface (gmsm_field2u.h): every stored value is in R = [0,4q) with normalised limbs, so
// the zero tests are exact without bound tracking. Used for the Fp2 groups (G2 of BN254 and BLS12-381).
template <class U, bool INL>
GMSM_HD void double_mixed_g(XYZZL<U> &acc, const U &px, const U &py) {
    const U Uu = lz_dbl(py);
    const U V = lz_sqr<INL>(Uu);
    const U W = lz_mul<INL>(Uu, V);
    const U S = lz_mul<INL>(px, V);
    const U XX = lz_sqr<INL>(px);
    const U M = lz_add(lz_dbl(XX), XX);
    const U X3 = lz_sub(lz_sqr<INL>(M), lz_dbl(S));
    const U Y3 = lz_sub(lz_mul<INL>(lz_sub(S, X3), M), lz_mul<INL>(W, py));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = V;
    acc.zzz = W;
}

template <class U, bool INL>
GMSM_HD void madd_g(XYZZL<U> &acc, bool &inf, const U &px, const U &py_in, bool negate) {
    const U py = negate ? lz_sub(lz_zero((const U *)nullptr), py_in) : py_in;
    if (inf) {
        acc.x = px;
        acc.y = py;
        acc.zz = lz_one((const U *)nullptr);
        acc.zzz = lz_one((const U *)nullptr);
        inf = false;
        return;
    }
    const U Pv = lz_sub(lz_mul<INL>(px, acc.zz), acc.x);
    const U Rv = lz_sub(lz_mul<INL>(py, acc.zzz), acc.y);
    if (lz_is_zero(Pv)) {  // g2.go: same special cases as g1.go:846-854
        if (lz_is_zero(Rv)) double_mixed_g<U, INL>(acc, px, py);
        else inf = true;
        return;
    }
    const U PP = lz_sqr<INL>(Pv);
    const U PPP = lz_mul<INL>(Pv, PP);
    const U Q = lz_mul<INL>(acc.x, PP);
    const U X3 = lz_sub(lz_sub(lz_sqr<INL>(Rv), PPP), lz_dbl(Q));
    const U Y3 = lz_sub(lz_mul<INL>(lz_sub(Q, X3), Rv), lz_mul<INL>(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = lz_mul<INL>(acc.zz, PP);
    acc.zzz = lz_mul<INL>(acc.zzz, PPP);
}

template <class U, bool INL>
GMSM_HD XYZZL<U> double_g(const XYZZL<U> &q) {
    const U Uu = lz_dbl(q.y);
    const U V = lz_sqr<INL>(Uu);
    const U W = lz_mul<INL>(Uu, V);
    const U S = lz_mul<INL>(q.x, V);
    const U XX = lz_sqr<INL>(q.x);
    const U M = lz_add(lz_dbl(XX), XX);
    XYZZL<U> r;
    r.x = lz_sub(lz_sqr<INL>(M), lz_dbl(S));
    r.y = lz_sub(lz_mul<INL>(lz_sub(S, r.x), M), lz_mul<INL>(W, q.y));
    r.zz = lz_mul<INL>(V, q.zz);
    r.zzz = lz_mul<INL>(W, q.zzz);
    return r;
}

template <class U, bool INL>
GMSM_HD void add_g(XYZZL<U> &p, bool &pinf, const XYZZL<U> &q, bool qinf) {
    if (qinf) return;
    if (pinf) {
        p = q;
        pinf = false;
        return;
    }
    const U U2 = lz_mul<INL>(q.x, p.zz);
    const U U1 = lz_mul<INL>(p.x, q.zz);
    const U S2 = lz_mul<INL>(q.y, p.zzz);
    const U S1 = lz_mul<INL>(p.y, q.zzz);
    const U A = lz_sub(U2, U1);
    const U B = lz_sub(S2, S1);
    if (lz_is_zero(A)) {
        if (lz_is_zero(B)) p = double_g<U, INL>(q);
        else pinf = true;
        return;
    }
    const U PP = lz_sqr<INL>(A);
    const U PPP = lz_mul<INL>(A, PP);
    const U Q = lz_mul<INL>(U1, PP);
    const U V = lz_mul<INL>(S1, PPP);
    const U X3 = lz_sub(lz_sub(lz_sqr<INL>(B), PPP), lz_dbl(Q));
    p.y = lz_sub(lz_mul<INL>(lz_sub(Q, X3), B), V);
    p.x = X3;
    p.zz = lz_mul<INL>(lz_mul<INL>(p.zz, q.zz), PP);
    p.zzz = lz_mul<INL>(lz_mul<INL>(p.zzz, q.zzz), PPP);
}

// ------------------------------------------------------------------ which mixed addition the accumulation loops run
// Both families run on SIGNED limbs: madd_s (prime fields) and madd_ts (Fp2), below. The unsigned forms above - madd_u,
// madd_g - stay as what the host checks compare them with (tests/c/lazy_signed_check.cpp) and for the few additions
// outside the loops (ingest). (Rounds 2-4 also carried a bound-tracked Fp2 form, madd_t, for BLS12-381 G2 and compile-time
// switches between all five; madd_ts replaced it for both Fp2 groups - profiles/r04_fp2_signed_ab.log - and it is gone.)
template <class U> struct LzSigned { static constexpr bool value = false; };
template <class P> struct LzSigned<FpU<P>> { static constexpr bool value = true; };
template <class P> struct LzSigned<Fp2U<P>> { static constexpr bool value = true; };

// back to the reduced class R (exactly normalised, < 4q) before a record leaves the accumulation loop
template <class P>
GMSM_HD void fpu_to_class_r(FpU<P> &a) {  // a < 12q, nearly normalised
    fpu_normalize(a);
    fpu_cond_sub_4q(a);
    fpu_cond_sub_4q(a);
}

// ------------------------------------------------------------------ the Fp2 mixed addition on signed limbs
// madd_s carried over to Fp2 = Fp[u]/(u^2 + 1): components are signed numbers on signed limbs, a difference is one
// subtraction per limb, a product component is ONE signed two-product scan (c0 = a0 b0 + (-a1) b1, c1 = a0 b1 + a1 b0: a
// negation is one instruction per limb, where the unsigned forms pay K q - a1 and a carry pass), and nothing is ever
// compared with q inside the loop. This needs no spare bits beyond the 7 of BN254's radix:
// without K q offsets the values stay within +-8q, so BN254 G2 leaves the reduced class R of madd_g - whose every
// subtraction is two sequential borrow chains and a conditional +4q - as well.
// Values in multiples of q, per component (A = 2^(L W)/q >= 169; m(.) = largest |component|; a product component lies in
// (-(m m' + m m')/A, (m m' + m m')/A + 1)):
//   px, py: m < 2;  accumulator: m(X) < 5.3, m(Y) < 2.5, m(ZZ), m(ZZZ) < 1.1  (or class R, < 4, after the doubling branch)
//   P = px ZZ - X: m < 7.2    R = py ZZZ - Y: m < 5.1
//   PP = P^2: a0 = (P0 + P1)(P0 - P1) in (-1.3, 2.3), a1 = 2 P0 P1 in (-0.7, 2.7);   RR likewise, smaller
//   PPP = P PP, Q = X PP: m < 1.3      X3 = RR - PPP - 2Q: m < 5.3      D = Q - X3: m < 6.6
//   Y3 = D R - Y PPP: one four-product scan per component where the columns hold 5L products (BLS12-381), else two
//   two-product scans, summed and carried (BN254: 27 products of < 2^58.1 per column is what a signed 64-bit column holds)
// Limbs: every operand of a scan is a scan result (normalised, signed top limb), carry-passed, or the difference of two
// such values: within +-(2^W + 2^(32-W)). Of a square, the sum and the difference of the components and the doubled cross
// product are carry-passed for that reason. P == 0 is tested on P^2 (a field: P^2 = 0 <=> P = 0), whose a0 can come out as
// -q, 0, q or 2q (fps_prod_is_zero). The accumulator leaves the class through lz_acc_finish / lz_rec_fresh: + 8q, then
// exactly into R.
template <class P>
GMSM_HD bool fps_prod_is_zero(const FpU<P> &a) {  // a == 0 mod q for a scan result in (-2q, 3q)
    const uint32_t l0 = a.l[0];
    if (l0 != 0u && l0 != P::UQ1[0] && l0 != P::UQ2[0] && l0 != ((0u - P::UQ1[0]) & FpU<P>::MASK)) return false;
    FpU<P> n = a;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) n.l[i] += P::UQ1[i];  // (-q, 4q)
    fps_normalize(n);
    return fpu_is_zero_r(n);
}
template <class P>
GMSM_HD Fp2U<P> f2s_sub(const Fp2U<P> &a, const Fp2U<P> &b) { return Fp2U<P>{fps_sub<P>(a.a0, b.a0), fps_sub<P>(a.a1, b.a1)}; }
template <class P>
GMSM_HD Fp2U<P> f2s_neg(const Fp2U<P> &a) { return Fp2U<P>{fps_neg<P>(a.a0), fps_neg<P>(a.a1)}; }
template <bool INL, class P>
GMSM_HD Fp2U<P> f2s_mul(const Fp2U<P> &x, const Fp2U<P> &y) {
    Fp2U<P> z;
    z.a0 = fsmuladd<INL>(x.a0, y.a0, fps_neg<P>(x.a1), y.a1);
    z.a1 = fsmuladd<INL>(x.a0, y.a1, x.a1, y.a0);
    return z;
}
// x^2: a0 = (x0 + x1)(x0 - x1), a1 = 2 x0 x1 (e2_bn254.go:41-50); `half1` = x0 x1 (for the zero test)
template <bool INL, class P>
GMSM_HD Fp2U<P> f2s_sqr(const Fp2U<P> &x, FpU<P> &half1) {
    FpU<P> sum = fps_add<P>(x.a0, x.a1), dif = fps_sub<P>(x.a0, x.a1);
    fps_carry(sum);
    fps_carry(dif);
    Fp2U<P> z;
    z.a0 = fsmul<INL>(sum, dif);
    half1 = fsmul<INL>(x.a0, x.a1);
    z.a1 = fps_add<P>(half1, half1);
    fps_carry(z.a1);
    return z;
}

template <class P, bool INL>
GMSM_HD void madd_ts(XYZZL<Fp2U<P>> &acc, bool &inf, const Fp2U<P> &px, const Fp2U<P> &py_in, bool negate) {
    using U = Fp2U<P>;
    const U py = negate ? f2s_neg<P>(py_in) : py_in;
    if (inf) {
        acc.x = px;
        acc.y = py;
        acc.zz = lz_one((const U *)nullptr);
        acc.zzz = lz_one((const U *)nullptr);
        inf = false;
        return;
    }
    const U Pv = f2s_sub<P>(f2s_mul<INL>(px, acc.zz), acc.x);
    const U Rv = f2s_sub<P>(f2s_mul<INL>(py, acc.zzz), acc.y);
    FpU<P> hp, hr;
    const U PP = f2s_sqr<INL>(Pv, hp);
    if (fps_prod_is_zero(PP.a0) && fps_prod_is_zero(hp)) {            // same x (g2.go, as g1.go:846-854)
        const U RR0 = f2s_sqr<INL>(Rv, hr);
        if (fps_prod_is_zero(RR0.a0) && fps_prod_is_zero(hr))         // P + P: the reduced-class doubling (rare)
            double_mixed_g<U, INL>(acc, px, negate ? lz_sub(lz_zero((const U *)nullptr), py_in) : py_in);
        else inf = true;                                              // P + (-P)
        return;
    }
    const U PPP = f2s_mul<INL>(Pv, PP);
    const U Q = f2s_mul<INL>(acc.x, PP);
    const U RR = f2s_sqr<INL>(Rv, hr);
    const U X3{fps_sub_sub2<P>(RR.a0, PPP.a0, Q.a0), fps_sub_sub2<P>(RR.a1, PPP.a1, Q.a1)};
    const U D = f2s_sub<P>(Q, X3);
    // Y3 = D R - Y PPP:  a0 = D0 R0 - D1 R1 - Y0 PPP0 + Y1 PPP1,  a1 = D0 R1 + D1 R0 - Y0 PPP1 - Y1 PPP0
    const FpU<P> nD1 = fps_neg<P>(D.a1), nY0 = fps_neg<P>(acc.y.a0), nY1 = fps_neg<P>(acc.y.a1);
    U Y3;
    if constexpr (INL && FpsFits4<P>::value) {
        Y3.a0 = fpu_mul_add4<P, true>(D.a0, Rv.a0, nD1, Rv.a1, nY0, PPP.a0, acc.y.a1, PPP.a1);
        Y3.a1 = fpu_mul_add4<P, true>(D.a0, Rv.a1, D.a1, Rv.a0, nY0, PPP.a1, nY1, PPP.a0);
    } else {
        Y3.a0 = fps_add<P>(fsmuladd<INL>(D.a0, Rv.a0, nD1, Rv.a1), fsmuladd<INL>(nY0, PPP.a0, acc.y.a1, PPP.a1));
        Y3.a1 = fps_add<P>(fsmuladd<INL>(D.a0, Rv.a1, D.a1, Rv.a0), fsmuladd<INL>(nY0, PPP.a1, nY1, PPP.a0));
        fps_carry(Y3.a0);
        fps_carry(Y3.a1);
    }
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f2s_mul<INL>(acc.zz, PP);
    acc.zzz = f2s_mul<INL>(acc.zzz, PPP);
}

// a signed component > -8q (carry-passed or a scan result) -> R
template <class P>
GMSM_HD void fps_to_class_r(FpU<P> &a) {
#pragma unroll
    for (int i = 0; i < P::UL; ++i) a.l[i] += fpu_kq<P, 8>(i);
    fpu_to_class_r(a);
}
template <class P>
GMSM_HD void f2s_to_class_r(Fp2U<P> &a) {
    fps_to_class_r(a.a0);
    fps_to_class_r(a.a1);
}


// ---- dispatch: prime-field elements use the bound-tracked forms above, Fp2 elements the reduced-class forms ----
template <bool INL, class P>
GMSM_HD void lz_madd(XYZZL<FpU<P>> &acc, bool &inf, const FpU<P> &px, const FpU<P> &py, bool negate) {
    madd_u<P, INL>(acc, inf, px, py, negate);
}
template <bool INL, class P>
GMSM_HD void lz_madd(XYZZL<Fp2U<P>> &acc, bool &inf, const Fp2U<P> &px, const Fp2U<P> &py, bool negate) {
    madd_g<Fp2U<P>, INL>(acc, inf, px, py, negate);
}
// a record k_accumulate_seg wrote (signed class of madd_s) -> the unsigned class every other kernel computes in: + 4q on
// the coordinates that can be negative. Harmless on a record that already is in the unsigned class (a bucket closed by
// the fix-up, the running buckets of a multi-range call): coordinates only enter the additions as operands of products,
// whose bounds have room for it (x, y < 11 + 8, zzz < 6 + 8: 19 * 14 / 169 + 1 < 4 on BN254, the narrowest field).
// (Fp2: + 8q and exactly into R = [0, 4q) on every component - R stays R.)
template <class U>
GMSM_HD void lz_coord_fresh(U &c, uint32_t which) {  // one coordinate (a lane of a quad): 0 x, 1 y, 2 zz, 3 zzz
    if constexpr (LzSigned<U>::value) {
        if constexpr (IsLazyPrimeField<U>::value) {
            if (which != 2u) fps_to_unsigned(c);  // zz is a product of non-negative values: already unsigned
        } else {
            f2s_to_class_r(c);
        }
    }
}
template <class U>
GMSM_HD void lz_rec_fresh(XYZZL<U> &v) {
    lz_coord_fresh(v.x, 0u);
    lz_coord_fresh(v.y, 1u);
    lz_coord_fresh(v.zz, 2u);
    lz_coord_fresh(v.zzz, 3u);
}
// the mixed addition of an accumulation LOOP (k_accumulate_seg, the fixed-base walk): may leave the coordinates in a
// wider class than the records in memory use; lz_acc_finish brings them back before the store
template <bool INL, class U>
GMSM_HD void lz_madd_acc(XYZZL<U> &acc, bool &inf, const U &px, const U &py, bool negate) {
    static_assert(LzSigned<U>::value, "the accumulation loops run on signed limbs");
    if constexpr (IsLazyPrimeField<U>::value) madd_s<typename U::Params, INL>(acc, inf, px, py, negate);
    else madd_ts<typename U::Params, INL>(acc, inf, px, py, negate);
}
// TO_RECORD: the value goes to a bucket / partial-sum record whose readers apply lz_rec_fresh (k_accumulate_seg: the
// flush sits on the divergent bucket-boundary path of the hot loop, executed by the whole wave for the one or two lanes
// whose bucket ends - 87 % of the iterations at 32 entries per bucket -, the readers run it once per record: measured
// -1.1 % with the conversion inside the loop's flush, profiles/r04_signed_limbs.log).
template <bool TO_RECORD = false, class U>
GMSM_HD void lz_acc_finish(XYZZL<U> &acc, bool inf) {
    if constexpr (!TO_RECORD) {
        if (!inf) lz_rec_fresh(acc);
    }
}
template <bool INL, class P>
GMSM_HD void lz_padd(XYZZL<FpU<P>> &p, bool &pinf, const XYZZL<FpU<P>> &q, bool qinf) { add_u<P, INL>(p, pinf, q, qinf); }
template <bool INL, class P>
GMSM_HD void lz_padd(XYZZL<Fp2U<P>> &p, bool &pinf, const XYZZL<Fp2U<P>> &q, bool qinf) { add_g<Fp2U<P>, INL>(p, pinf, q, qinf); }
template <bool INL, class P>
GMSM_HD XYZZL<FpU<P>> lz_pdbl(const XYZZL<FpU<P>> &q) { return double_u<P, INL>(q); }
template <bool INL, class P>
GMSM_HD XYZZL<Fp2U<P>> lz_pdbl(const XYZZL<Fp2U<P>> &q) { return double_g<Fp2U<P>, INL>(q); }

#if defined(__HIPCC__)  // everything below touches device memory (the group law above also compiles for the host: tests/c)
// ------------------------------------------------------------------ arithmetic policies
template <class T>
__device__ __forceinline__ T policy_load(const void *base, size_t index) {
    static_assert(sizeof(T) % 16 == 0, "element size");
    T r;
    const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(base) + index * sizeof(T));
    uint4 *dst = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 16); ++i) dst[i] = src[i];
    return r;
}

template <class T>
__device__ __forceinline__ void policy_store(void *base, size_t index, const T &v) {
    static_assert(sizeof(T) % 16 == 0, "element size");
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(base) + index * sizeof(T));
    const uint4 *src = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 16); ++i) dst[i] = src[i];
}

template <class U>
struct UnsatElem {
    XYZZL<U> v;
    bool inf;
};

// Bucket / partial-sum records of the lazy path live in HBM in the lazy representation itself (4 x element limbs, no
// conversion on either side): infinity <=> every limb of zz is zero (a finite point has zz != 0 mod q, so its limbs
// cannot all vanish). Only the per-window totals leave the device and are converted to canonical saturated limbs.
template <class U>
GMSM_HD bool lz_limbs_all_zero(const U &a) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&a);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(U) / 4); ++i) acc |= w[i];
    return acc == 0;
}

template <class U>
__device__ __forceinline__ UnsatElem<U> unsat_load(const void *base, size_t i) {
    UnsatElem<U> e;
    e.v = policy_load<XYZZL<U>>(base, i);
    e.inf = lz_limbs_all_zero(e.v.zz);
    return e;
}

// a record written by k_accumulate_seg (partial sums always, buckets unless the fix-up closed them)
template <class U>
__device__ __forceinline__ UnsatElem<U> unsat_load_fresh(const void *base, size_t i) {
    UnsatElem<U> e = unsat_load<U>(base, i);
    if (!e.inf) lz_rec_fresh(e.v);
    return e;
}

template <class U>
__device__ __forceinline__ void lazy_store(void *base, size_t i, const XYZZL<U> &v, bool inf) {
    // x, y, zzz go out as they are; zz is ANDed with an all-ones / all-zero mask (infinity <=> zz limbs all zero): no
    // copy of the 4 x sizeof(U) record is built in registers - this sits on the bucket-boundary path of the hot loop
    static_assert(sizeof(U) % 4 == 0 && sizeof(XYZZL<U>) % 16 == 0, "record layout");
    constexpr int UW = (int)(sizeof(U) / 4);
    const uint32_t keep = inf ? 0u : 0xffffffffu;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(base) + i * sizeof(XYZZL<U>));
#pragma unroll
    for (int q = 0; q < (int)(sizeof(XYZZL<U>) / 16); ++q) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int word = 4 * q + j;
            o[j] = (word >= 2 * UW && word < 3 * UW) ? (w[word] & keep) : w[word];
        }
        dst[q] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// canonical saturated XYZZ for the host (window totals)
template <class U, bool INL>
__device__ __forceinline__ void unsat_store_final(void *base, size_t i, const UnsatElem<U> &e) {
    using T = LzTraits<U>;
    using Mem = XYZZ<typename T::Sat>;
    Mem m = Mem::infinity();
    if (!e.inf) {
        m.x = T::template to_sat<INL>(e.v.x);
        m.y = T::template to_sat<INL>(e.v.y);
        m.zz = T::template to_sat<INL>(e.v.zz);
        m.zzz = T::template to_sat<INL>(e.v.zzz);
    }
    policy_store<Mem>(base, i, m);
}

template <class U>
__device__ __forceinline__ UnsatElem<U> unsat_infinity() {
    UnsatElem<U> e;
    e.inf = true;
    e.v.x = e.v.y = lz_one((const U *)nullptr);
    e.v.zz = e.v.zzz = lz_one((const U *)nullptr);
    return e;
}

// Fully inlined policy: for kernels with ONE call site of the addition (k_fixup_seg, k_reduce_serial).
template <class U>
struct UnsatOps {
    using Mem = XYZZL<U>;                              // bucket / partial record in HBM
    using Final = XYZZ<typename LzTraits<U>::Sat>;     // window total handed to the host
    using Elem = UnsatElem<U>;
    __device__ static __forceinline__ Elem infinity() { return unsat_infinity<U>(); }
    __device__ static __forceinline__ Elem load(const void *base, size_t i) { return unsat_load<U>(base, i); }
    __device__ static __forceinline__ Elem load_fresh(const void *base, size_t i) { return unsat_load_fresh<U>(base, i); }
    __device__ static __forceinline__ void fresh(Elem &e) {  // what load_fresh does, for a record that was loaded raw
        if (!e.inf) lz_rec_fresh(e.v);
    }
    __device__ static __forceinline__ void store(void *base, size_t i, const Elem &e) { lazy_store<U>(base, i, e.v, e.inf); }
    __device__ static __forceinline__ void store_final(void *base, size_t i, const Elem &e) { unsat_store_final<U, true>(base, i, e); }
    __device__ static __forceinline__ void add(Elem &p, const Elem &q) { lz_padd<true>(p.v, p.inf, q.v, q.inf); }
    __device__ static __forceinline__ void dbl(Elem &p) {
        if (!p.inf) p.v = lz_pdbl<true>(p.v);
    }
};

#endif  // __HIPCC__

}  // namespace gmsm
