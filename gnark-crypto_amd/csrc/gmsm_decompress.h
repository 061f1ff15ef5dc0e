// Compressed point encoding on the device (the Encoder's DEFAULT format - what kzg.SRS.WriteTo / ReadFrom
// move, ecc/bn254/kzg/marshal.go:20-63): X only, with the flag bits saying which of the two roots Y is.
//
//   k_decompress  n compressed points -> Go-layout affine points: (*G1Affine).setBytes, compressed branch
//                 (ecc/bn254/marshal.go:907-948; G2 :1168-1215; the slice form the Decoder runs in parallel:
//                 unsafeSetCompressedBytes + unsafeComputeY, :951-1027): X canonical, Y^2 = X^3 + b, Y = sqrt(Y^2) or
//                 "invalid compressed coordinate: square root doesn't exist", Y or -Y by LexicographicallyLargest
//                 (fp/element.go:282-296; E2: A1 unless it is zero, internal/fptower/e2.go:46-52) against the flag
//   k_compress    the inverse: (*G1Affine).Bytes (marshal.go:801-823, G2 :1051-1075)
//
// Flags (top bits of the first byte; gmsm_params32.h COMP_*): BN254 10 smallest / 11 largest / 01 infinity
// (marshal.go:26-30); BLS12-381 and BW6-761 100 / 101 / 110 (bls12-381/marshal.go:25-35). Fp2: X.A1 | X.A0.
//
// The square root. All three base fields are 3 mod 4: w = a^((q-3)/4) gives sqrt(a) = w a AND 1/sqrt(a) = w (w * w a =
// a^((q-1)/2) = 1 for a residue), one exponentiation of ~BITS squarings on the lazy limbs of the MSM pipeline. Over Fp2 =
// Fp[u]/(u^2 + 1) the reference exponentiates IN Fp2 twice (E2.Sqrt, e2.go:211-234: algorithm 9 of eprint 2012/685); the
// result is one of the two roots and the flag picks between them, so any method that finds a root decodes to the same point.
// Here: the norm N = a0^2 + a1^2 must be a square s^2 (E2.Legendre is the Legendre symbol of the norm, e2.go:155-159: the
// reference's "square root doesn't exist"), then with t = (a0 +- s) / 2 - exactly one of the two is a residue, their product
// is -(a1/2)^2 - the root is sqrt(t) + a1 / (2 sqrt(t)) u: two or three exponentiations in Fp (a third of an Fp2 one each).
#pragma once
#include "gmsm_ingest.h"

namespace gmsm {

// x^((q-3)/4) for q = 3 mod 4: the exponent is q >> 2, its bit j is bit j + 2 of q. Left-to-right with a sliding window of
// three bits over the odd powers x, x^3, x^5, x^7: every squaring of the binary method and about a quarter as many products
// as the exponent has bits (BN254: 251 squarings + 58 products instead of + 108, BLS12-381 378 + 108 instead of + 227, BW6-761
// 759 + 179 instead of + 343). The exponent is a constant, so the
// control flow is the same in every lane; the one squaring and the one product of the loop are inlined (two bodies).
// Operands in the product routine's input range; result < 2q.
template <class P>
__device__ __noinline__ FpU<P> fpu_pow_q4(const FpU<P> &x) {
    static_assert((P::Q[0] & 3u) == 3u, "sqrt by one exponentiation needs q = 3 mod 4");
    auto bit = [](int j) -> uint32_t { return (P::Q[(j + 2) >> 5] >> ((j + 2) & 31)) & 1u; };  // of the exponent q >> 2
    const FpU<P> x2 = fsqr<false>(x);  // the four odd powers: one-off products, out of line
    const FpU<P> p3 = fmul<false>(x, x2), p5 = fmul<false>(p3, x2), p7 = fmul<false>(p5, x2);
    FpU<P> r = x;
    bool started = false;
    int mulpos = -1;  // where the window that is being squared through ends (its lowest bit, which is set)
    uint32_t val = 0;
#pragma nounroll
    for (int j = P::BITS - 3; j >= 0; --j) {  // from the exponent's top bit down: ONE squaring site, ONE product site
        if (mulpos < 0 && bit(j)) {           // a window of at most three bits opens at a set bit and ends in one
            mulpos = j >= 2 ? j - 2 : 0;
            while (!bit(mulpos)) ++mulpos;
            val = 0;
            for (int k = j; k >= mulpos; --k) val = (val << 1) | bit(k);
        }
        if (started) r = fsqr<true>(r);
        if (j == mulpos) {
            FpU<P> m;
#pragma unroll
            for (int i = 0; i < P::UL; ++i) {  // the odd power `val`, selected limb by limb (the condition is uniform)
                const uint32_t a01 = (val & 2u) ? p3.l[i] : x.l[i], a23 = (val & 2u) ? p7.l[i] : p5.l[i];
                m.l[i] = (val & 4u) ? a23 : a01;
            }
            r = started ? fmul<true>(r, m) : m;
            started = true;
            mulpos = -1;
        }
    }
    return r;
}

template <class P>
__device__ __forceinline__ bool fpu_equal_mod_q(const FpU<P> &a, const FpU<P> &b) {  // both of the reduced class
    return fpu_is_zero_r(fpu_subr(a, b));
}

// y = sqrt(a) when it exists (then *inv_y = 1/y unless a = 0); a of the reduced class
template <class P>
__device__ __forceinline__ bool fpu_sqrt(const FpU<P> &a, FpU<P> &y, FpU<P> *inv_y = nullptr) {
    const FpU<P> w = fpu_pow_q4(a);
    y = fmul<false>(w, a);
    if (inv_y) *inv_y = w;
    return fpu_equal_mod_q(fsqr<false>(y), a);
}

template <class P>
__device__ __forceinline__ bool lz_sqrt(const FpU<P> &a, FpU<P> &y) {
    return fpu_sqrt(a, y);
}
template <class P>
__device__ bool lz_sqrt(const Fp2U<P> &a, Fp2U<P> &y) {
    const FpU<P> zero = lz_zero((const FpU<P> *)nullptr);
    if (fpu_is_zero_r(a.a1)) {  // a in Fp: sqrt(a0), or sqrt(-a0) u (-1 is not a square)
        y.a1 = zero;
        if (fpu_sqrt(a.a0, y.a0)) return true;
        y.a0 = zero;
        return fpu_sqrt(fpu_subr(zero, a.a0), y.a1);
    }
    FpU<P> s, x0, w;
    Fp<P> h;
#pragma unroll
    for (int i = 0; i < P::N; ++i) h.l[i] = P::INV2[i];
    const FpU<P> half = fpu_from_sat<P, false>(h);
    const FpU<P> norm = fpu_addr(fsqr<false>(a.a0), fsqr<false>(a.a1));
    if (!fpu_sqrt(norm, s)) return false;
    FpU<P> t = fmul<false>(fpu_addr(a.a0, s), half);
    if (!fpu_sqrt(t, x0, &w)) {
        t = fmul<false>(fpu_subr(a.a0, s), half);
        if (!fpu_sqrt(t, x0, &w)) return false;  // cannot happen: t+ t- = -(a1/2)^2 is a non-residue
    }
    y.a0 = x0;
    y.a1 = fmul<false>(fmul<false>(a.a1, w), half);
    return true;
}

// LexicographicallyLargest of a canonical Montgomery element: its regular form is >= (q + 1) / 2
template <class P>
__device__ bool fp_lex_largest(const Fp<P> &mont) {
    const Fp<P> z = fp_from_mont(mont);
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const uint64_t d = (uint64_t)z.l[i] - P::LEX_HALF[i] - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return borrow == 0;
}
template <class P>
__device__ bool fp_lex_largest(const Fp2<P> &mont) {
    return mont.a1.is_zero() ? fp_lex_largest(mont.a0) : fp_lex_largest(mont.a1);
}

// canonical Montgomery element -> big-endian bytes of its regular form (BigEndian.PutElement, fp/element.go)
template <class P>
__device__ void fp_to_be(const Fp<P> &mont, uint8_t *dst) {
    const Fp<P> z = fp_from_mont(mont);
    uint32_t *w = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
    for (int i = 0; i < P::N; ++i) w[P::N - 1 - i] = __builtin_bswap32(z.l[i]);
}
template <class P>
__device__ void coord_to_be(const Fp<P> &x, uint8_t *dst) { fp_to_be(x, dst); }
template <class P>
__device__ void coord_to_be(const Fp2<P> &x, uint8_t *dst) {  // A1 first
    fp_to_be(x.a1, dst);
    fp_to_be(x.a0, dst + 4 * P::N);
}

// One thread per point. comp: n * sizeof(F) bytes; out: Go-layout affine points, offenders written as infinity and reported
// (first offender wins, as in k_decode_raw). The subgroup check is the caller's next launch (k_validate_points).
template <class F, class C>
__global__ void __launch_bounds__(128) k_decompress(const uint8_t *__restrict__ comp, size_t n, Affine<F> *__restrict__ out,
                                                    unsigned long long *first_bad) {
    using U = typename IngestLazy<F>::type;
    using T = LzTraits<U>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int CB = (int)sizeof(F);  // bytes of a compressed point = one coordinate
    const uint8_t *src = comp + i * CB;
    constexpr uint8_t FLAG_MASK = (uint8_t)(0xff << (8 - C::RAW_FLAG_BITS));
    const uint8_t flag = (uint8_t)((src[0] & FLAG_MASK) >> (8 - C::RAW_FLAG_BITS));
    Affine<F> a{F::zero(), F::zero()};
    uint32_t status = PT_OK;
    if (flag == (uint8_t)C::COMP_INFINITY) {
        uint32_t acc = src[0] & (uint8_t)~FLAG_MASK;  // the rest of the buffer must be zero (isZeroed, marshal.go:433)
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
        for (int k = 1; k < CB / 4; ++k) acc |= w[k];
        acc |= w[0] & 0xffffff00u;
        if (acc != 0) status = PT_BAD_INFINITY;
    } else if (flag != (uint8_t)C::COMP_SMALLEST && flag != (uint8_t)C::COMP_LARGEST) {
        status = PT_BAD_FLAG;  // an uncompressed or undefined flag: this entry takes Bytes() output
    } else if (!coord_from_be(src, (uint8_t)~FLAG_MASK, a.x)) {
        status = PT_NOT_CANONICAL;
    } else {
        const U x = T::template from_sat<false>(a.x);
        const U rhs = lz_add(lz_mul<false>(lz_sqr<false>(x), x), T::template from_sat<false>(coeff_b<typename F::Params, C>((const F *)nullptr)));
        U y;
        if (!lz_sqrt(rhs, y)) {
            status = PT_NO_SQRT;
        } else {
            a.y = T::template to_sat<false>(y);
            if (fp_lex_largest(a.y) != (flag == (uint8_t)C::COMP_LARGEST)) a.y = fp_neg(a.y);
        }
    }
    if (status != PT_OK) {
        report_bad(first_bad, i, status);
        a = Affine<F>{F::zero(), F::zero()};
    }
    out[i] = a;
}

// (*G1Affine).Bytes: infinity = the infinity flag over zeroes; else X with the flag of Y's half
template <class F, class C>
__global__ void __launch_bounds__(128) k_compress(const Affine<F> *__restrict__ pts, size_t n, uint8_t *__restrict__ comp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int CB = (int)sizeof(F);
    uint8_t *dst = comp + i * CB;
    const Affine<F> a = pts[i];
    if (a.is_infinity()) {
        uint32_t *w = reinterpret_cast<uint32_t *>(dst);
        for (int k = 0; k < CB / 4; ++k) w[k] = 0;
        dst[0] = (uint8_t)(C::COMP_INFINITY << (8 - C::RAW_FLAG_BITS));
        return;
    }
    coord_to_be(a.x, dst);
    dst[0] |= (uint8_t)((fp_lex_largest(a.y) ? C::COMP_LARGEST : C::COMP_SMALLEST) << (8 - C::RAW_FLAG_BITS));
}

}  // namespace gmsm
