// Point ingest: the step before the MSM in a prover pipeline - an SRS or a vector of points
// arrives as bytes and has to become validated Montgomery limbs in HBM.
//
//   k_decode_raw      n points in the reference's uncompressed wire format -> Go-layout affine points (Montgomery)
//                     (*G1Affine).RawBytes / setBytes, ecc/bn254/marshal.go:826-846, :862-905 (G2 :1078-1112);
//                     the 3-flag-bit variant of BLS12-381 / BW6-761: ecc/bls12-381/marshal.go:27-34, :855-880, :901-945
//   k_validate_points the checks the reference's Decoder runs over a slice (marshal.go:250-275): on the curve and in the
//                     r-torsion (IsInSubGroup, g1.go:190, :478; bls12-381/g1.go:481; g2.go twins)
//
// Wire format (big-endian, REGULAR form - PutElement leaves the Montgomery domain, fp/element.go BigEndian.PutElement):
//   G1: X | Y;   G2 over Fp2: X.A1 | X.A0 | Y.A1 | Y.A0.   Metadata sits in the top RAW_FLAG_BITS of the first byte.
//   BN254: 2 bits, 00 = uncompressed; the point at infinity is 64 (128) zero bytes under that same flag.
//   BLS12-381, BW6-761: 3 bits, 000 = uncompressed, 010 = uncompressed infinity (every other bit must be zero).
//   Compressed encodings (Bytes()) are refused here: this is the RawBytes / RawEncoding() ingest path; gmsm_decompress.h
//   takes those.
//
// Subgroup membership (check level 2) is decided by the reference's endomorphism identities (gmsm_subgroup.h: [x^2] phi(P) + P
// etc., 2-6 x fewer group operations than the definition); level 3 decides it as the definition says - [r]P = infinity -,
// the same predicate for every point ON the curve, kept as the cross-check of level 2 (rounds 1-5 ran it for level 2).
// Points that are not on the curve are always rejected, whatever the subgroup flag says for the rest.
#pragma once
#include <hip/hip_runtime.h>
#include "gmsm_context.h"
#include "gmsm_curve.h"
#include "gmsm_curveu.h"
#include "gmsm_subgroup.h"

namespace gmsm {

// element of the coordinate field from its group constant (Fp: N words; Fp2: a0 then a1)
template <class P, class C>
GMSM_HD Fp<P> coeff_b(const Fp<P> *) {
    Fp<P> b;
#pragma unroll
    for (int i = 0; i < P::N; ++i) b.l[i] = C::B[i];
    return b;
}
template <class P, class C>
GMSM_HD Fp2<P> coeff_b(const Fp2<P> *) {
    Fp2<P> b;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        b.a0.l[i] = C::B[i];
        b.a1.l[i] = C::B[P::N + i];
    }
    return b;
}

template <class P>
GMSM_HD bool limbs_equal(const Fp<P> &a, const Fp<P> &b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) acc |= a.l[i] ^ b.l[i];
    return acc == 0;
}
template <class P>
GMSM_HD bool limbs_equal(const Fp2<P> &a, const Fp2<P> &b) { return limbs_equal(a.a0, b.a0) && limbs_equal(a.a1, b.a1); }

// y^2 == x^3 + b; the point at infinity (0,0) counts as on the curve (G1Affine.IsOnCurve goes through FromAffine, g1.go:183)
template <class F, class C>
GMSM_HD bool point_on_curve(const Affine<F> &a) {
    if (a.is_infinity()) return true;
    const F lhs = fp_sqr(a.y);
    const F rhs = fp_add(fp_mul(fp_sqr(a.x), a.x), coeff_b<typename F::Params, C>((const F *)nullptr));
    return limbs_equal(lhs, rhs);
}

// [r]P == infinity by double-and-add over the bits of the scalar-field modulus - on the lazy limbs of the MSM pipeline
// (gmsm_curveu.h) since round 4: the saturated group law (gmsm_curve.h: a carry chain per partial product) ran this loop
// at 31 G products/s for BLS12-381 G1, the lazy one runs the accumulation kernel at twice that. Same special cases: the
// doubling of a 2-torsion point (zz becomes 0 mod q: tested after every doubling, exactly) and P + (-P) (inside the
// mixed addition) both lead to the infinity flag.
template <class F, class FrP>
__device__ bool point_in_r_torsion(const Affine<F> &a) {
    using U = typename IngestLazy<F>::type;
    using T = LzTraits<U>;
    constexpr bool INL = sizeof(U) <= 14 * 4;  // wider elements call their products (code size, as in gmsm_fixedbase.h)
    if (a.is_infinity()) return true;
    const U px = T::template from_sat<INL>(a.x), py = T::template from_sat<INL>(a.y);
    XYZZL<U> acc;
    bool inf = true;
#pragma nounroll
    for (int bit = FrP::BITS - 1; bit >= 0; --bit) {
        if (!inf) {
            acc = lz_pdbl<INL>(acc);
            if (ingest_zz_is_zero(acc.zz)) inf = true;
        }
        if ((FrP::Q[bit >> 5] >> (bit & 31)) & 1u) lz_madd<INL>(acc, inf, px, py, false);
    }
    return inf;
}

// level 0: nothing, 1: on the curve, 2: on the curve and in the r-torsion by the reference's endomorphism identity, 3: the
// same decided by [r]P = infinity (BY_DEF) (NEEDS_TORSION false: prime-order curve, the curve check is the subgroup check - BN254 G1,
// g1.go:475-482)
// BY_DEF picks the subgroup test at COMPILE time (level 3 launches the BY_DEF kernels): one kernel holding both loops is sized - registers
// and call frames - for the larger of the two, which is the 255/377-bit walk of the definition (round 6).
template <class F, class FrP, class C, bool NEEDS_TORSION, bool BY_DEF>
__device__ uint32_t validate_point(const Affine<F> &a, int level) {
    if (level <= 0) return PT_OK;
    if (!point_on_curve<F, C>(a)) return PT_NOT_ON_CURVE;
    if constexpr (NEEDS_TORSION) {
        if (level >= 2 && !a.is_infinity()) {
            bool in;
            if constexpr (BY_DEF) in = point_in_r_torsion<F, FrP>(a);
            else in = point_in_subgroup_endo<F, C>(a);
            if (!in) return PT_NOT_IN_SUBGROUP;
        }
    }
    return PT_OK;
}

// first_bad: (index << 3 | status), the smallest over all offending points; ~0 when every point passed
__device__ __forceinline__ void report_bad(unsigned long long *first_bad, size_t i, uint32_t status) {
    atomicMin(first_bad, ((unsigned long long)i << 3) | status);
}

// big-endian bytes of one Fp -> canonical saturated words; false when the value is not below q.
// `src` is 4*N bytes, 4-byte aligned; `first_byte_mask` clears the metadata bits of the very first byte.
template <class P>
__device__ bool fp_from_be(const uint8_t *src, uint8_t first_byte_mask, Fp<P> &out) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t v = __builtin_bswap32(w[P::N - 1 - i]);
        if (i == P::N - 1) v &= ((uint32_t)first_byte_mask << 24) | 0x00ffffffu;
        out.l[i] = v;
    }
    bool lt = false, eq = true;  // lexicographic from the top word (smallerThanModulus, fp/element.go:345)
#pragma unroll
    for (int i = P::N - 1; i >= 0; --i) {
        lt = lt || (eq && out.l[i] < P::Q[i]);
        eq = eq && out.l[i] == P::Q[i];
    }
    return lt;
}

template <class P>
__device__ Fp<P> fp_to_mont(const Fp<P> &x) {
    Fp<P> rsq;
#pragma unroll
    for (int i = 0; i < P::N; ++i) rsq.l[i] = P::RSQ[i];
    return fp_mul(x, rsq);
}

template <class P>
__device__ bool coord_from_be(const uint8_t *src, uint8_t first_byte_mask, Fp<P> &out) {
    Fp<P> t;
    if (!fp_from_be<P>(src, first_byte_mask, t)) return false;
    out = fp_to_mont(t);
    return true;
}
template <class P>
__device__ bool coord_from_be(const uint8_t *src, uint8_t first_byte_mask, Fp2<P> &out) {  // A1 first, then A0
    Fp<P> t1, t0;
    if (!fp_from_be<P>(src, first_byte_mask, t1)) return false;
    if (!fp_from_be<P>(src + 4 * P::N, 0xff, t0)) return false;
    out.a1 = fp_to_mont(t1);
    out.a0 = fp_to_mont(t0);
    return true;
}

// One thread per point. raw: n * sizeof(Affine<F>) bytes (the uncompressed size equals the in-memory size for every
// group in scope); out: Go-layout affine points. Offending points are written as infinity and reported.
template <class F, class FrP, class C, bool NEEDS_TORSION, bool BY_DEF>
__global__ void __launch_bounds__(128) k_decode_raw(const uint8_t *__restrict__ raw, size_t n, int level,
                                                    Affine<F> *__restrict__ out, unsigned long long *first_bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int BYTES = (int)sizeof(Affine<F>);
    const uint8_t *src = raw + i * BYTES;
    constexpr uint8_t FLAG_MASK = (uint8_t)(0xff << (8 - C::RAW_FLAG_BITS));
    const uint8_t flag = (uint8_t)((src[0] & FLAG_MASK) >> (8 - C::RAW_FLAG_BITS));
    Affine<F> a{F::zero(), F::zero()};
    uint32_t status = PT_OK;
    if (C::RAW_INFINITY_FLAG >= 0 && flag == (uint8_t)C::RAW_INFINITY_FLAG) {
        // the rest of the buffer must be zero (isZeroed, marshal.go:433)
        uint32_t acc = src[0] & (uint8_t)~FLAG_MASK;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
        for (int k = 1; k < BYTES / 4; ++k) acc |= w[k];
        acc |= w[0] & 0xffffff00u;  // bytes 1..3 of the first word (memory order: byte 0 is the low byte)
        if (acc != 0) status = PT_BAD_INFINITY;
    } else if (flag != 0) {
        status = PT_BAD_FLAG;
    } else {
        constexpr int CB = BYTES / 2;  // bytes per coordinate
        if (!coord_from_be(src, (uint8_t)~FLAG_MASK, a.x) || !coord_from_be(src + CB, 0xff, a.y)) {
            status = PT_NOT_CANONICAL;
            a = Affine<F>{F::zero(), F::zero()};
        } else {
            status = validate_point<F, FrP, C, NEEDS_TORSION, BY_DEF>(a, level);
        }
    }
    if (status != PT_OK) {
        report_bad(first_bad, i, status);
        a = Affine<F>{F::zero(), F::zero()};
    }
    out[i] = a;
}

// The same checks over points that are already Montgomery limbs (an SRS dump is raw memory, utils/unsafe/dump_slice.go;
// ReadDump itself validates nothing, this is the optional check after it).
template <class F, class FrP, class C, bool NEEDS_TORSION, bool BY_DEF>
__global__ void __launch_bounds__(128) k_validate_points(const Affine<F> *__restrict__ pts, size_t n, int level,
                                                         unsigned long long *first_bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t status = validate_point<F, FrP, C, NEEDS_TORSION, BY_DEF>(pts[i], level);
    if (status != PT_OK) report_bad(first_bad, i, status);
}

}  // namespace gmsm
