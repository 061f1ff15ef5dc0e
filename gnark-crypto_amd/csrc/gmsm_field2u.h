// Reduced-class lazy arithmetic and the quadratic extension on top of the unsaturated limbs of gmsm_fieldu.h.
//
// The prime-field fast path (gmsm_curveu.h, madd_u/add_u) tracks per-formula bounds and never compares with q. That
// does not carry over to Fp2 = Fp[u]/(u^2+1): Karatsuba adds and subtracts three products per multiplication and, with
// only 7 spare bits in BN254's 2^261 radix, the bounds would run away. The G2 path therefore keeps every stored value
// in the *reduced class* R = [0, 4q) with exactly normalised limbs: additions and subtractions end with one
// conditional +-4q, products come out < 2q by themselves (operands up to 8q are fine: 64 q^2 / 2^261 < 0.4 q).
// (The accumulation LOOP of both G2 groups left R in round 4: on signed limbs nothing needs a K q offset, the values stay
// within +-8q, and 7 spare bits are enough - gmsm_curveu.h madd_ts; its records return to R when they are read. Fixup and
// reduction work on R throughout. Before: madd_g on R for BN254, a bound-tracked form (11 spare bits) for BLS12-381.)
//
// Replaces: fptower.E2 Add/Sub/Double/Neg/Mul/Square (ecc/bn254/internal/fptower/e2_fallback.go:10-28,
// e2_bn254.go:28-50; BLS12-381: e2_bls381.go:15-38). Exact arithmetic mod q: converting back with f2u_to_sat gives the
// canonical limbs the reference holds.
#pragma once
#include "gmsm_fieldu.h"

namespace gmsm {

// t normalised, t < 8q  ->  t - 4q if t >= 4q
template <class P>
GMSM_HD void fpu_cond_sub_4q(FpU<P> &a) {
    constexpr int L = P::UL;
    uint32_t d[L];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t t = a.l[i] - P::UQ4[i] - borrow;
        borrow = t >> 31;  // limbs < 2^31: bit 31 set <=> negative
        d[i] = (i < L - 1) ? (t & FpU<P>::MASK) : t;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) a.l[i] = borrow ? a.l[i] : d[i];
}

// R x R -> R
template <class P>
GMSM_HD FpU<P> fpu_addr(const FpU<P> &a, const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + b.l[i];
    fpu_normalize(r);
    fpu_cond_sub_4q(r);
    return r;
}

// R x R -> R : a - b, plus 4q when negative. Sequential signed borrow chain (limbs are exactly normalised).
template <class P>
GMSM_HD FpU<P> fpu_subr(const FpU<P> &a, const FpU<P> &b) {
    constexpr int L = P::UL, W = P::UW;
    FpU<P> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t t = a.l[i] - b.l[i] - borrow;
        borrow = t >> 31;
        r.l[i] = (i < L - 1) ? (t & FpU<P>::MASK) : t;
    }
    // borrow out of the top limb <=> a < b: add 4q (two's complement top limb wraps back)
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t t = r.l[i] + (borrow ? P::UQ4[i] : 0u) + c;
        if (i < L - 1) {
            r.l[i] = t & FpU<P>::MASK;
            c = t >> W;
        } else {
            r.l[i] = t;
        }
    }
    return r;
}

template <class P>
GMSM_HD FpU<P> fpu_dblr(const FpU<P> &a) {
    return fpu_addr(a, a);
}

// unreduced sum / difference used only as multiplier operands (< 8q, limbs <= 2^(W+1))
template <class P>
GMSM_HD FpU<P> fpu_add_raw(const FpU<P> &a, const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class P>
GMSM_HD bool fpu_is_zero_r(const FpU<P> &a) {  // a in R, normalised: 0, q, 2q, 3q
    uint32_t z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) {
        z0 |= a.l[i];
        z1 |= a.l[i] ^ P::UQ1[i];
        z2 |= a.l[i] ^ P::UQ2[i];
        z3 |= a.l[i] ^ P::UQ3[i];
    }
    return z0 == 0 || z1 == 0 || z2 == 0 || z3 == 0;
}

// ------------------------------------------------------------------ Fp2 on lazy limbs
template <class P>
struct Fp2U {
    using Params = P;
    FpU<P> a0, a1;
};

template <class U> struct IsLazyPrimeField { static constexpr bool value = false; };
template <class P> struct IsLazyPrimeField<FpU<P>> { static constexpr bool value = true; };

// ---- the "lz_" interface the generic group law is written against (both element types) ----
template <bool INL, class P> GMSM_HD FpU<P> lz_mul(const FpU<P> &a, const FpU<P> &b) { return fmul<INL>(a, b); }
template <bool INL, class P> GMSM_HD FpU<P> lz_sqr(const FpU<P> &a) { return fsqr<INL>(a); }
template <class P> GMSM_HD FpU<P> lz_add(const FpU<P> &a, const FpU<P> &b) { return fpu_addr(a, b); }
template <class P> GMSM_HD FpU<P> lz_sub(const FpU<P> &a, const FpU<P> &b) { return fpu_subr(a, b); }
template <class P> GMSM_HD FpU<P> lz_dbl(const FpU<P> &a) { return fpu_dblr(a); }
template <class P> GMSM_HD bool lz_is_zero(const FpU<P> &a) { return fpu_is_zero_r(a); }
template <class P> GMSM_HD FpU<P> lz_zero(const FpU<P> *) {
    FpU<P> z;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) z.l[i] = 0;
    return z;
}
template <class P> GMSM_HD FpU<P> lz_one(const FpU<P> *) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = P::UONE[i];
    return r;
}

// (a0 + a1 u)(b0 + b1 u), u^2 = -1 (e2_bn254.go:28-37), as two double products with ONE Montgomery reduction each:
//   c0 = a0 b0 + (8q - a1) b1,   c1 = a0 b1 + a1 b0        (operands in R = [0,4q): 48 q^2 / 2^(L W) + q < 2q, in R;
//   8q and not 4q: a1 may be as large as 4q - 1, one unit above the top limb of the borrow form of 4q)
// Same number of multiplies as Karatsuba's three reduced products (6 L^2 + 2 L against 6 L^2 + 3 L) but none of its
// three exact subtractions, each a sequential borrow chain with a conditional +4q (measured in round 2: BLS12-381 G2
// accumulates ~20 % slower with the Karatsuba form; tests/c/lazy_field_check.cpp keeps one as the reference).
template <bool INL, class P>
GMSM_HD Fp2U<P> lz_mul(const Fp2U<P> &x, const Fp2U<P> &y) {
    Fp2U<P> z;
    z.a0 = fmuladd<INL>(x.a0, y.a0, fpu_neg8n<P>(x.a1), y.a1);
    z.a1 = fmuladd<INL>(x.a0, y.a1, x.a1, y.a0);
    return z;
}

template <bool INL, class P>
GMSM_HD Fp2U<P> lz_sqr(const Fp2U<P> &x) {  // e2_bn254.go:41-50
    const FpU<P> s = fpu_add_raw(x.a0, x.a1);   // < 8q
    const FpU<P> d = fpu_subr(x.a0, x.a1);      // R
    Fp2U<P> z;
    z.a0 = fmul<INL>(s, d);
    z.a1 = fpu_dblr(fmul<INL>(x.a0, x.a1));
    return z;
}

template <class P> GMSM_HD Fp2U<P> lz_add(const Fp2U<P> &a, const Fp2U<P> &b) { return Fp2U<P>{fpu_addr(a.a0, b.a0), fpu_addr(a.a1, b.a1)}; }
template <class P> GMSM_HD Fp2U<P> lz_sub(const Fp2U<P> &a, const Fp2U<P> &b) { return Fp2U<P>{fpu_subr(a.a0, b.a0), fpu_subr(a.a1, b.a1)}; }
template <class P> GMSM_HD Fp2U<P> lz_dbl(const Fp2U<P> &a) { return Fp2U<P>{fpu_dblr(a.a0), fpu_dblr(a.a1)}; }
template <class P> GMSM_HD bool lz_is_zero(const Fp2U<P> &a) { return fpu_is_zero_r(a.a0) && fpu_is_zero_r(a.a1); }
template <class P> GMSM_HD Fp2U<P> lz_zero(const Fp2U<P> *) { return Fp2U<P>{lz_zero((const FpU<P> *)nullptr), lz_zero((const FpU<P> *)nullptr)}; }
template <class P> GMSM_HD Fp2U<P> lz_one(const Fp2U<P> *) { return Fp2U<P>{lz_one((const FpU<P> *)nullptr), lz_zero((const FpU<P> *)nullptr)}; }

// ---- element <-> saturated / packed forms ----
template <class U> struct LzTraits;
template <class P>
struct LzTraits<FpU<P>> {
    using Params = P;
    using Sat = Fp<P>;
    static constexpr int PACKED_WORDS = P::N;  // one coordinate, packed lazy-domain words
    template <bool INL> GMSM_HD static FpU<P> from_sat(const Fp<P> &x) { return fpu_from_sat<P, INL>(x); }
    template <bool INL> GMSM_HD static Fp<P> to_sat(const FpU<P> &x) { return fpu_to_sat<P, INL>(x); }
    GMSM_HD static FpU<P> unpack(const uint32_t *w) { return fpu_unpack<P>(w); }
    GMSM_HD static void pack(const FpU<P> &x, uint32_t *w) { fpu_pack(x, w); }
};
template <class P>
struct LzTraits<Fp2U<P>> {
    using Params = P;
    using Sat = Fp2<P>;
    static constexpr int PACKED_WORDS = 2 * P::N;
    template <bool INL> GMSM_HD static Fp2U<P> from_sat(const Fp2<P> &x) {
        return Fp2U<P>{fpu_from_sat<P, INL>(x.a0), fpu_from_sat<P, INL>(x.a1)};
    }
    template <bool INL> GMSM_HD static Fp2<P> to_sat(const Fp2U<P> &x) {
        return Fp2<P>{fpu_to_sat<P, INL>(x.a0), fpu_to_sat<P, INL>(x.a1)};
    }
    GMSM_HD static Fp2U<P> unpack(const uint32_t *w) { return Fp2U<P>{fpu_unpack<P>(w), fpu_unpack<P>(w + P::N)}; }
    GMSM_HD static void pack(const Fp2U<P> &x, uint32_t *w) {
        fpu_pack(x.a0, w);
        fpu_pack(x.a1, w + P::N);
    }
};

}  // namespace gmsm
