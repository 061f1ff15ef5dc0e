// Extended-Jacobian group law on the unsaturated ("lazy") field representation of gmsm_fieldu.h, plus the two
// arithmetic policies the bucket kernels are written against:
//   SatOps<F>    -- generic saturated arithmetic (gmsm_curve.h), any coordinate field (Fp or Fp2)
//   UnsatOps<P>  -- unsaturated arithmetic for groups with coordinates in Fp (all G1, BW6-761 G2)
// Both present the same interface (Elem, load/store of the canonical saturated XYZZ record in memory, add, dbl), so the
// fixup and reduction kernels are written once.
//
// Same group semantics as the reference's g1JacExtended (ecc/bn254/g1.go:682-985): every special case is kept
// (inf + P, P + inf, P + P -> doubling, P + (-P) -> infinity). Value bounds are stated in multiples of q and hold for
// every field in scope because 2^(UL*UW)/q >= 169 (BN254 261-254 = 7 spare bits; BLS12-381 392-381 = 11; BW6-761 23):
//   stored coordinates: x < 11, y < 7, zz, zzz < 3;   mul(a,b) < a*b/169 + 1.
#pragma once
#include "gmsm_curve.h"
#include "gmsm_fieldu.h"

namespace gmsm {

template <class P>
struct XYZZU {
    FpU<P> x, y, zz, zzz;
};

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
    const FpU<P> X3 = fpu_sub<P, 4>(fpu_sub<P, 4>(RR, PPP), fpu_dbl(Q));                       // < 11
    const FpU<P> Y3 = fpu_sub<P, 4>(fmul<INL>(fpu_sub<P, 16>(Q, X3), Rv), fmul<INL>(acc.y, PPP));  // < 7
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fmul<INL>(acc.zz, PP);
    acc.zzz = fmul<INL>(acc.zzz, PPP);
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
    const FpU<P> V = fmul<INL>(S1, PPP);                                  // < 2
    const FpU<P> X3 = fpu_sub<P, 4>(fpu_sub<P, 4>(fsqr<INL>(B), PPP), fpu_dbl(Q));  // < 2 + 4 + 4
    p.y = fpu_sub<P, 4>(fmul<INL>(fpu_sub<P, 16>(Q, X3), B), V);          // < 6
    p.x = X3;
    p.zz = fmul<INL>(fmul<INL>(p.zz, q.zz), PP);
    p.zzz = fmul<INL>(fmul<INL>(p.zzz, q.zzz), PPP);
}

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

template <class F>
struct SatOps {
    using Field = F;
    using Mem = XYZZ<F>;  // record in HBM: canonical saturated Montgomery, infinity <=> zz == 0
    struct Elem {
        XYZZ<F> v;
    };
    __device__ static __forceinline__ Elem infinity() { return Elem{XYZZ<F>::infinity()}; }
    __device__ static __forceinline__ Elem load(const void *base, size_t i) { return Elem{policy_load<Mem>(base, i)}; }
    __device__ static __forceinline__ void store(void *base, size_t i, const Elem &e) { policy_store<Mem>(base, i, e.v); }
    __device__ static __forceinline__ void add(Elem &p, const Elem &q) { xyzz_add(p.v, q.v); }
    __device__ static __forceinline__ void dbl(Elem &p) { p.v = xyzz_double(p.v); }
};

template <class P>
struct UnsatElem {
    XYZZU<P> v;
    bool inf;
};

template <class P, bool INL>
__device__ __forceinline__ UnsatElem<P> unsat_load(const void *base, size_t i) {
    using Mem = XYZZ<Fp<P>>;
    const Mem m = policy_load<Mem>(base, i);
    UnsatElem<P> e;
    e.inf = m.zz.is_zero();
    e.v.x = fpu_from_sat<P, INL>(m.x);
    e.v.y = fpu_from_sat<P, INL>(m.y);
    e.v.zz = fpu_from_sat<P, INL>(m.zz);
    e.v.zzz = fpu_from_sat<P, INL>(m.zzz);
    return e;
}

template <class P, bool INL>
__device__ __forceinline__ void unsat_store(void *base, size_t i, const UnsatElem<P> &e) {
    using Mem = XYZZ<Fp<P>>;
    Mem m = Mem::infinity();
    if (!e.inf) {
        m.x = fpu_to_sat<P, INL>(e.v.x);
        m.y = fpu_to_sat<P, INL>(e.v.y);
        m.zz = fpu_to_sat<P, INL>(e.v.zz);
        m.zzz = fpu_to_sat<P, INL>(e.v.zzz);
    }
    policy_store<Mem>(base, i, m);
}

template <class P>
__device__ __forceinline__ UnsatElem<P> unsat_infinity() {
    UnsatElem<P> e;
    e.inf = true;
    e.v.x = e.v.y = fpu_one<P>();
    e.v.zz = e.v.zzz = fpu_one<P>();
    return e;
}

// Fully inlined policy: fastest when the kernel holds few call sites (k_fixup_seg, k_reduce1/2).
template <class P>
struct UnsatOps {
    using Field = Fp<P>;
    using Mem = XYZZ<Fp<P>>;
    using Elem = UnsatElem<P>;
    __device__ static __forceinline__ Elem infinity() { return unsat_infinity<P>(); }
    __device__ static __forceinline__ Elem load(const void *base, size_t i) { return unsat_load<P, true>(base, i); }
    __device__ static __forceinline__ void store(void *base, size_t i, const Elem &e) { unsat_store<P, true>(base, i, e); }
    __device__ static __forceinline__ void add(Elem &p, const Elem &q) { add_u<P, true>(p.v, p.inf, q.v, q.inf); }
    __device__ static __forceinline__ void dbl(Elem &p) {
        if (!p.inf) p.v = double_u<P, true>(p.v);
    }
};

// Out-of-line policy on the out-of-line multiplier: smallest code. k_fixup_level has five call sites of the group
// operations inside one loop and ran 15x slower fully inlined (instruction-cache overflow).
template <class P>
struct UnsatOpsNI {
    using Field = Fp<P>;
    using Mem = XYZZ<Fp<P>>;
    using Elem = UnsatElem<P>;
    __device__ static __forceinline__ Elem infinity() { return unsat_infinity<P>(); }
    __device__ static __noinline__ Elem load(const void *base, size_t i) { return unsat_load<P, false>(base, i); }
    __device__ static __noinline__ void store(void *base, size_t i, const Elem &e) { unsat_store<P, false>(base, i, e); }
    __device__ static __noinline__ void add(Elem &p, const Elem &q) { add_u<P, false>(p.v, p.inf, q.v, q.inf); }
    __device__ static __noinline__ void dbl(Elem &p) {
        if (!p.inf) p.v = double_u<P, false>(p.v);
    }
};

}  // namespace gmsm
