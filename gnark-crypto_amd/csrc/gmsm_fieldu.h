// Unsaturated ("lazy") prime-field arithmetic for the bucket-accumulation hot loop on gfx950.
//
// Why a second representation: measured on MI355X (tools/ubench_valu.hip, profiles/r01_valu_issue_rates.log)
// v_mad_u64_u32 (32x32+64 -> 64) issues every 4 cycles per wave64, the same as v_add_co/v_addc_co, and it has a
// 64-bit addend but no carry-in.  With saturated 32-bit limbs every product needs a second carry instruction and the
// compiler adds ~2 register moves to build the 64-bit addend pair (r01a: 128 mads + 105 v_lshl_add_u64 + 320 v_mov per
// field multiplication).  With UL limbs of UW < 32 bits, a whole column of the product (2*UL partial products) fits one
// 64-bit accumulator, so the Montgomery product is product-scanning with exactly ONE v_mad_u64_u32 per partial
// product, one 64-bit shift per column, and no carry chains:  (2*UL^2 + UL) multiplies vs 2*N^2 + N with ~3x the
// instructions around them.
//
// Representation: value = sum l[i] * 2^(UW*i), i < UL; "nearly normalised" limbs l[i] <= 2^UW + 2^3 (top limb free);
// values are NOT reduced modulo q after every step: each op documents the bound (in multiples of q) it needs and gives.
// Montgomery radix 2^(UL*UW) (BN254: 9 x 29 = 261 bits, 7 spare bits over q), so products of operands up to ~13q
// still come out below 2q..3q without any conditional subtraction.
//
// Bit-exactness: these are exact integer computations modulo q; fpu_to_sat() returns the canonical (fully reduced)
// saturated Montgomery limbs the reference would hold, so results are compared limb-for-limb with the oracle.
//
// Replaces in the reference (same role, different shape): fp.Element Mul/Square/Add/Sub/Double/Neg,
// ecc/bn254/fp/element.go:386-454 and element_purego.go:46-213 (asm: field/asm/element_4w_amd64.s:208-304).
#pragma once
#include "gmsm_field.h"

namespace gmsm {

template <class P>
struct FpU {
    using Params = P;
    static constexpr int L = P::UL;
    static constexpr int W = P::UW;
    static constexpr uint32_t MASK = (1u << P::UW) - 1u;
    uint32_t l[L];
};

// One-level carry pass: limbs <= 2^32-1 in, limbs <= 2^W + 2^(32-W) out (top limb absorbs its carry). Value unchanged.
template <class P>
GMSM_HD void fpu_carry(FpU<P> &a) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    uint32_t c[L];
#pragma unroll
    for (int i = 0; i < L - 1; ++i) c[i] = a.l[i] >> W;
#pragma unroll
    for (int i = L - 1; i >= 1; --i) a.l[i] = (i < L - 1 ? (a.l[i] & MASK) : a.l[i]) + c[i - 1];
    a.l[0] &= MASK;
}

// a + b. Value bound: bound(a) + bound(b). Inputs nearly normalised.
template <class P>
GMSM_HD FpU<P> fpu_add(const FpU<P> &a, const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + b.l[i];
    fpu_carry(r);
    return r;
}

template <class P>
GMSM_HD FpU<P> fpu_dbl(const FpU<P> &a) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] << 1;
    fpu_carry(r);
    return r;
}

// the redundant form of K*q (gmsm_params32.h: every low limb carries 4*2^W borrowed from the limb above)
template <class P, int K>
GMSM_HD constexpr uint32_t fpu_kq(int i) {
    static_assert(K == 4 || K == 8 || K == 16 || K == 32, "redundant multiples of q in the parameter tables");
    return K == 4 ? P::UK4[i] : K == 8 ? P::UK8[i] : K == 16 ? P::UK16[i] : P::UK32[i];
}
// a - b + K*q with K = 4, 8, 16 or 32: requires b < K*q (strictly: top limb of b <= top limb of the redundant K*q) and
// b's limbs < 2^(W+2). Value bound: bound(a) + K.
template <class P, int K>
GMSM_HD FpU<P> fpu_sub(const FpU<P> &a, const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + fpu_kq<P, K>(i) - b.l[i];
    fpu_carry(r);
    return r;
}
// K*q - b, carry-passed (nearly normalised limbs, fit to be any operand of fpu_mul_add); same requirement on b
template <class P, int K>
GMSM_HD FpU<P> fpu_negc(const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = fpu_kq<P, K>(i) - b.l[i];
    fpu_carry(r);
    return r;
}

// 4q - b for a normalised b < 2q (the negation used for subMixed). UK4N borrows one unit per limb, so every result limb
// is in (0, 2^(W+1)) and can feed the multiplier directly: no carry pass.
template <class P>
GMSM_HD FpU<P> fpu_neg4(const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = P::UK4N[i] - b.l[i];
    return r;
}

// 8q - b for a normalised b < 4q (any member of the reduced class R of gmsm_field2u.h): like fpu_neg4 but safe up to
// 4q - 1, where 4q's own top limb would underflow. Result limbs in (0, 2^(W+1)), no carry pass.
template <class P>
GMSM_HD FpU<P> fpu_neg8n(const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = P::UK8N[i] - b.l[i];
    return r;
}

// a - b - 2c + 8q in one pass (X3 = R^2 - PPP - 2Q): requires b + 2c < 8q limb-wise (b, c nearly normalised).
// Value bound: bound(a) + 8.
template <class P>
GMSM_HD FpU<P> fpu_sub_sub2(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + P::UK8[i] - b.l[i] - (c.l[i] << 1);
    fpu_carry(r);
    return r;
}

// Column accumulators of the product scans, unsigned (every kernel but one) or signed (the accumulation loop's mixed
// addition, "signed limbs" below): same instruction count, v_mad_i64_i32 instead of v_mad_u64_u32 and an arithmetic
// column shift.
template <bool SIGNED>
struct LimbAcc {
    using type = uint64_t;
    static GMSM_HD type mul(uint32_t a, uint32_t b) { return (uint64_t)a * b; }
};
template <>
struct LimbAcc<true> {
    using type = int64_t;
    static GMSM_HD type mul(uint32_t a, uint32_t b) { return (int64_t)(int32_t)a * (int64_t)(int32_t)b; }
};

// Independent accumulator chains per product column (GMSM_MUL_NACC, set per translation unit = per group).
// The product scan adds every partial product of a column into ONE 64-bit accumulator: a dependency chain of up to
// 2*UL v_mad_u64_u32. Kernels that run several waves per SIMD hide it; the wide element types (28-limb BW6-761, Fp2 over
// 14-limb BLS12-381) only fit ONE wave per SIMD and their register pressure leaves the compiler no room to interleave
// independent columns (ISA: one chain through a single register pair), so that lone wave issues at the multiplier's
// result latency instead of its issue rate. With NACC chains the partial products of a column alternate between NACC
// accumulators that are summed once per column: NACC-1 extra 64-bit additions per column (3.5 % of the instructions at
// NACC = 2, UL = 28).
#ifndef GMSM_MUL_NACC
#define GMSM_MUL_NACC 1
#endif

// Montgomery product a*b*2^-(L*W) mod q, product scanning. Requires limbs of a, b <= 2^(W+1) and
// bound(a)*bound(b) <= 2^(L*W)/q * (B-1) for the output bound B (BN254: 2^261/q = 169, so 13q x 13q -> < 2q).
// Output limbs are normalised (< 2^W), top limb holds the rest.
template <class P, bool SIGNED = false>
GMSM_HD FpU<P> fpu_mul(const FpU<P> &a, const FpU<P> &b) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    constexpr int NACC = GMSM_MUL_NACC;
    uint32_t m[L];
    FpU<P> r;
    using LA = LimbAcc<SIGNED>;
    using A = typename LA::type;
    A acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        A part[NACC > 1 ? NACC - 1 : 1] = {0};  // side chains of this column (chain 0 is `acc`, which carries over)
        int t = 0;
        const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
        for (int i = lo; i <= hi; ++i, ++t) {
            const A pr = LA::mul(a.l[i], b.l[k - i]);
            if (NACC > 1 && (t % NACC) != 0) part[(t % NACC) - 1] += pr;
            else acc += pr;
        }
        // reduction row: m[i] * q[k-i] for the m's known so far (i < k, i < L) with q index k-i in [1, L)
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i < (k < L ? k : L); ++i, ++t) {
            const A pr = LA::mul(m[i], P::UQ[k - i]);
            if (NACC > 1 && (t % NACC) != 0) part[(t % NACC) - 1] += pr;
            else acc += pr;
        }
        if (NACC > 1) {
#pragma unroll
            for (int j = 0; j < NACC - 1; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm("" : "+v"(part[j]));  // opaque to the reassociation pass, which would fold the chains back into one
#endif
                acc += part[j];
            }
        }
        if (k < L) {
            m[k] = ((uint32_t)acc * P::UQINV) & MASK;
            acc += LA::mul(m[k], P::UQ[0]);
        } else {
            r.l[k - L] = (uint32_t)acc & MASK;
        }
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

// (a*b + c*d) * 2^-(L*W) mod q with ONE Montgomery reduction: both product rows go into the same column accumulators
// (3L products of < 2^(2W+0.1) per column: 27 * 2^58.1 < 2^63 for L = 9, W = 29 - the operands must be nearly normalised,
// limbs <= 2^W + 2^(32-W), which every carry-passed value is). Saves the L^2 + L multiplies of a second reduction.
// Bound: (bound(a) bound(b) + bound(c) bound(d)) / (2^(L*W)/q) + 1.
template <class P, bool SIGNED = false>
GMSM_HD FpU<P> fpu_mul_add(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c, const FpU<P> &d) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    // worst admitted mix: one operand of one product un-carried (fpu_neg4: limbs < 2^(W+1)), everything else nearly
    // normalised: L * (2^(2W+1) + 2 * 2^(2W)) per column
    static_assert((unsigned long long)L * 4 * (1ull << (2 * W)) < (1ull << 63) + ((1ull << 63) - 1), "column accumulator overflow");
    // signed limbs: every operand limb within +-(2^W + 2^(32-W)), 2L products and L reduction products per column
    static_assert(!SIGNED || 3ull * L * (((1ull << W) + (1ull << (32 - W))) * ((1ull << W) + (1ull << (32 - W)))) < (1ull << 63),
                  "signed column accumulator overflow");
    uint32_t m[L];
    FpU<P> r;
    using LA = LimbAcc<SIGNED>;
    using A = typename LA::type;
    A acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            acc += LA::mul(a.l[i], b.l[k - i]);
            acc += LA::mul(c.l[i], d.l[k - i]);
        }
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i < (k < L ? k : L); ++i) acc += LA::mul(m[i], P::UQ[k - i]);
        if (k < L) {
            m[k] = ((uint32_t)acc * P::UQINV) & MASK;
            acc += LA::mul(m[k], P::UQ[0]);
        } else {
            r.l[k - L] = (uint32_t)acc & MASK;
        }
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

// (a*b + c*d + e*f + g*h) * 2^-(L*W) mod q with ONE reduction: one component of the difference of two Fp2 products
// (gmsm_curveu.h, madd_ts). Every operand nearly normalised (limbs <= 2^W + 2^(32-W)): 5L products per column.
// Bound: (sum of the four bound products) / (2^(L*W)/q) + 1.
template <class P, bool SIGNED = false>
GMSM_HD FpU<P> fpu_mul_add4(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c, const FpU<P> &d, const FpU<P> &e,
                            const FpU<P> &f, const FpU<P> &g, const FpU<P> &h) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    static_assert((unsigned long long)L * 5 * ((1ull << (2 * W)) + (1ull << (W + 6))) < (1ull << 63), "column accumulator overflow");
    static_assert(!SIGNED || 5ull * L * (((1ull << W) + (1ull << (32 - W))) * ((1ull << W) + (1ull << (32 - W)))) < (1ull << 63),
                  "signed column accumulator overflow");
    uint32_t m[L];
    FpU<P> r;
    using LA = LimbAcc<SIGNED>;
    using A = typename LA::type;
    A acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            acc += LA::mul(a.l[i], b.l[k - i]);
            acc += LA::mul(c.l[i], d.l[k - i]);
            acc += LA::mul(e.l[i], f.l[k - i]);
            acc += LA::mul(g.l[i], h.l[k - i]);
        }
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i < (k < L ? k : L); ++i) acc += LA::mul(m[i], P::UQ[k - i]);
        if (k < L) {
            m[k] = ((uint32_t)acc * P::UQINV) & MASK;
            acc += LA::mul(m[k], P::UQ[0]);
        } else {
            r.l[k - L] = (uint32_t)acc & MASK;
        }
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

// K*q - b, carry-passed (nearly normalised limbs): the subtrahend of a merged product, K = 8 (b < 8q)
template <class P>
GMSM_HD FpU<P> fpu_neg8c(const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = P::UK8[i] - b.l[i];
    fpu_carry(r);
    return r;
}

// Montgomery square: cross products once with a doubled operand (L(L+1)/2 instead of L^2 products for a*a).
template <class P, bool SIGNED = false>
GMSM_HD FpU<P> fpu_sqr(const FpU<P> &a) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    constexpr int NACC = GMSM_MUL_NACC;
    uint32_t m[L], d[L];
#pragma unroll
    for (int i = 0; i < L; ++i) d[i] = a.l[i] << 1;
    FpU<P> r;
    using LA = LimbAcc<SIGNED>;
    using A = typename LA::type;
    A acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
        A part[NACC > 1 ? NACC - 1 : 1] = {0};
        int t = 0;
        const int lo = k < L ? 0 : k - L + 1;
#pragma unroll
        for (int i = lo; 2 * i < k; ++i, ++t) {
            const A pr = LA::mul(d[i], a.l[k - i]);
            if (NACC > 1 && (t % NACC) != 0) part[(t % NACC) - 1] += pr;
            else acc += pr;
        }
        if ((k & 1) == 0) {
            const A pr = LA::mul(a.l[k / 2], a.l[k / 2]);
            if (NACC > 1 && (t % NACC) != 0) part[(t % NACC) - 1] += pr;
            else acc += pr;
            ++t;
        }
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i < (k < L ? k : L); ++i, ++t) {
            const A pr = LA::mul(m[i], P::UQ[k - i]);
            if (NACC > 1 && (t % NACC) != 0) part[(t % NACC) - 1] += pr;
            else acc += pr;
        }
        if (NACC > 1) {
#pragma unroll
            for (int j = 0; j < NACC - 1; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm("" : "+v"(part[j]));  // opaque to the reassociation pass, which would fold the chains back into one
#endif
                acc += part[j];
            }
        }
        if (k < L) {
            m[k] = ((uint32_t)acc * P::UQINV) & MASK;
            acc += LA::mul(m[k], P::UQ[0]);
        } else {
            r.l[k - L] = (uint32_t)acc & MASK;
        }
        acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
}

// Out-of-line copies for the kernels whose instruction footprint matters more than a call (chain fixup, bucket
// reduction: a fully inlined XYZZ addition is ~25 KB of code and those kernels would otherwise hold several copies,
// overflowing the 64 KB instruction cache -- measured: 15x slower). The accumulation loop keeps the inlined forms.
template <class P>
__host__ __device__ __noinline__ FpU<P> fpu_mul_ni(FpU<P> a, FpU<P> b) {
    return fpu_mul(a, b);
}
template <class P>
__host__ __device__ __noinline__ FpU<P> fpu_sqr_ni(FpU<P> a) {
    return fpu_sqr(a);
}
template <bool INL, class P>
GMSM_HD FpU<P> fmul(const FpU<P> &a, const FpU<P> &b) {
    if constexpr (INL) return fpu_mul(a, b);
    else return fpu_mul_ni<P>(a, b);
}
template <class P>
__host__ __device__ __noinline__ FpU<P> fpu_mul_add_ni(FpU<P> a, FpU<P> b, FpU<P> c, FpU<P> d) {
    return fpu_mul_add(a, b, c, d);
}
template <bool INL, class P>
GMSM_HD FpU<P> fmuladd(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c, const FpU<P> &d) {
    if constexpr (INL) return fpu_mul_add(a, b, c, d);
    else return fpu_mul_add_ni<P>(a, b, c, d);
}
template <bool INL, class P>
GMSM_HD FpU<P> fsqr(const FpU<P> &a) {
    if constexpr (INL) return fpu_sqr(a);
    else return fpu_sqr_ni<P>(a);
}

// ------------------------------------------------------------------ signed limbs
// The mixed addition of the accumulation loop (gmsm_curveu.h, madd_s) keeps its intermediate values as SIGNED numbers
// with signed limbs: value = sum (int32)l[i] * 2^(W i). A difference is then one subtraction per limb - no K*q to keep
// it positive and no carry pass to make room for the K*q - and the products run on v_mad_i64_i32 with signed 64-bit
// column accumulators (fpu_mul<P, true> etc.): the Montgomery step  m = (column * -q^-1) mod 2^W  works on the low bits
// of a two's complement column unchanged, m*q is added, and the column shift is arithmetic. For |a*b| < A*R the result
// lies in (-A, A + q); its low limbs are normalised, [0, 2^W), and the top limb carries the sign.
// Operands: every limb within +-(2^W + 2^(32-W)) (top limb free), checked per formula at the call sites.
// Measured on BN254 G1 (k_accumulate_seg, 2^20): see profiles/r04_signed_limbs.log.
template <class P>
GMSM_HD FpU<P> fps_sub(const FpU<P> &a, const FpU<P> &b) {  // a - b, limb by limb
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
}
template <class P>
GMSM_HD FpU<P> fps_neg(const FpU<P> &b) {
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = 0u - b.l[i];
    return r;
}
// One-level signed carry pass: limbs within +-2^31 in, limbs in [-2^(31-W), 2^W + 2^(31-W)) out. Value unchanged.
template <class P>
GMSM_HD void fps_carry(FpU<P> &a) {
    constexpr int L = P::UL, W = P::UW;
    constexpr uint32_t MASK = FpU<P>::MASK;
    uint32_t c[L];
#pragma unroll
    for (int i = 0; i < L - 1; ++i) c[i] = (uint32_t)((int32_t)a.l[i] >> W);
#pragma unroll
    for (int i = L - 1; i >= 1; --i) a.l[i] = (i < L - 1 ? (a.l[i] & MASK) : a.l[i]) + c[i - 1];
    a.l[0] &= MASK;
}
template <class P>
GMSM_HD FpU<P> fps_add(const FpU<P> &a, const FpU<P> &b) {  // a + b, limb by limb (no carry)
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// exact sequential normalisation of a signed value: low limbs in [0, 2^W), the top limb carries the rest and the sign
template <class P>
GMSM_HD void fps_normalize(FpU<P> &a) {
#pragma unroll
    for (int i = 0; i < P::UL - 1; ++i) {
        const uint32_t c = (uint32_t)((int32_t)a.l[i] >> P::UW);
        a.l[i] &= FpU<P>::MASK;
        a.l[i + 1] += c;
    }
}
// can the column accumulators of a signed four-product scan (fpu_mul_add4<P, true>) hold 5L products?
template <class P>
struct FpsFits4 {
    static constexpr unsigned long long LIM = (1ull << P::UW) + (1ull << (32 - P::UW));
    static constexpr bool value = 5ull * P::UL * (LIM * LIM) < (1ull << 63);
};
// a - b - 2c, carry-passed (X3 = R^2 - PPP - 2Q): a, b, c with limbs in [0, 2^W] -> limbs within (-3 * 2^W, 2^W], W <= 29
template <class P>
GMSM_HD FpU<P> fps_sub_sub2(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c) {
    static_assert(P::UW <= 29, "three limbs and a doubling in an int32");
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) r.l[i] = a.l[i] - ((c.l[i] << 1) + b.l[i]);
    fps_carry(r);
    return r;
}
// signed value > -4q, carry-passed -> the same residue as an unsigned nearly normalised value (+ 4q)
template <class P>
GMSM_HD void fps_to_unsigned(FpU<P> &a) {
#pragma unroll
    for (int i = 0; i < P::UL; ++i) a.l[i] += fpu_kq<P, 4>(i);
    fpu_carry(a);
}
template <class P>
__host__ __device__ __noinline__ FpU<P> fps_mul_ni(FpU<P> a, FpU<P> b) {
    return fpu_mul<P, true>(a, b);
}
template <class P>
__host__ __device__ __noinline__ FpU<P> fps_sqr_ni(FpU<P> a) {
    return fpu_sqr<P, true>(a);
}
template <class P>
__host__ __device__ __noinline__ FpU<P> fps_mul_add_ni(FpU<P> a, FpU<P> b, FpU<P> c, FpU<P> d) {
    return fpu_mul_add<P, true>(a, b, c, d);
}
template <bool INL, class P>
GMSM_HD FpU<P> fsmul(const FpU<P> &a, const FpU<P> &b) {
    if constexpr (INL) return fpu_mul<P, true>(a, b);
    else return fps_mul_ni<P>(a, b);
}
template <bool INL, class P>
GMSM_HD FpU<P> fssqr(const FpU<P> &a) {
    if constexpr (INL) return fpu_sqr<P, true>(a);
    else return fps_sqr_ni<P>(a);
}
template <bool INL, class P>
GMSM_HD FpU<P> fsmuladd(const FpU<P> &a, const FpU<P> &b, const FpU<P> &c, const FpU<P> &d) {
    if constexpr (INL) return fpu_mul_add<P, true>(a, b, c, d);
    else return fps_mul_add_ni<P>(a, b, c, d);
}

// ------------------------------------------------------------------ conversions
// Split the N saturated 32-bit words of a value < 2^(L*W) into L limbs of W bits (no arithmetic).
template <class P>
GMSM_HD FpU<P> fpu_unpack(const uint32_t *w) {
    constexpr int L = P::UL, W = P::UW, N = P::N;
    FpU<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int bit = i * W, j = bit >> 5, s = bit & 31;
        uint32_t v = j < N ? (w[j] >> s) : 0u;
        if (s + W > 32 && j + 1 < N) v |= w[j + 1] << (32 - s);
        r.l[i] = i < L - 1 ? (v & FpU<P>::MASK) : v;
    }
    return r;
}

// Pack fully normalised limbs (value < 2^(32N)) into N saturated words.
template <class P>
GMSM_HD void fpu_pack(const FpU<P> &a, uint32_t *w) {
    constexpr int L = P::UL, W = P::UW, N = P::N;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // word j covers bits [32j, 32j+32)
        const int lo_limb = (32 * j) / W, s = 32 * j - lo_limb * W;
        uint32_t v = a.l[lo_limb] >> s;
        if (lo_limb + 1 < L) v |= a.l[lo_limb + 1] << (W - s);
        if (2 * W - s < 32 && lo_limb + 2 < L) v |= a.l[lo_limb + 2] << (2 * W - s);
        w[j] = v;
    }
}

// Exact sequential normalisation (all limbs < 2^W, top limb carries the rest).
template <class P>
GMSM_HD void fpu_normalize(FpU<P> &a) {
    constexpr int L = P::UL, W = P::UW;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        uint32_t t = a.l[i] + c;
        a.l[i] = t & FpU<P>::MASK;
        c = t >> W;
    }
    a.l[L - 1] += c;
}

// value >= q ? value - q : value, for a fully normalised value < 2q; result canonical.
template <class P>
GMSM_HD void fpu_cond_sub_q(FpU<P> &a) {
    constexpr int L = P::UL, W = P::UW;
    uint32_t d[L];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        uint32_t t = a.l[i] - P::UQ[i] - borrow;
        borrow = (i < L - 1) ? (t >> 31) : (t >> 31);  // limbs < 2^31, so bit 31 set <=> negative
        d[i] = (i < L - 1) ? (t & FpU<P>::MASK) : t;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) a.l[i] = borrow ? a.l[i] : d[i];
    (void)W;
}

// Saturated Montgomery (x*2^(32N), canonical) -> unsaturated Montgomery (x*2^(L*W)), value < 2q.
template <class P, bool INL = true>
GMSM_HD FpU<P> fpu_from_sat(const Fp<P> &x) {
    FpU<P> u = fpu_unpack<P>(x.l);
    FpU<P> c;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) c.l[i] = P::UCIN[i];
    return fmul<INL>(u, c);
}

// Unsaturated Montgomery (any lazy value within the fpu_mul input bounds) -> canonical saturated Montgomery limbs.
template <class P, bool INL = true>
GMSM_HD Fp<P> fpu_to_sat(const FpU<P> &a) {
    FpU<P> c;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) c.l[i] = P::UCOUT[i];
    FpU<P> r = fmul<INL>(a, c);  // < 2q, limbs normalised
    fpu_cond_sub_q(r);
    Fp<P> z;
    fpu_pack(r, z.l);
    return z;
}

// ------------------------------------------------------------------ helpers of the fr/fft butterflies (gmsm_fft_lazy.h)
template <class P>
GMSM_HD constexpr uint32_t fpu_k2q(int i) {  // redundant 2q: a + k2q - b has no negative limb for b < 2q - 2*2^(W(L-1))
    return P::UQ2[i] + (i < P::UL - 1 ? (2u << P::UW) : 0u) - (i > 0 ? 2u : 0u);
}
// value >= K q ? value - K q : value for a fully normalised value (K = 1, 2)
template <class P, int K>
GMSM_HD void fpu_cond_sub_kq(FpU<P> &a) {
    constexpr int L = P::UL;
    uint32_t d[L];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t t = a.l[i] - (K == 1 ? P::UQ1[i] : P::UQ2[i]) - borrow;
        borrow = t >> 31;
        d[i] = (i < L - 1) ? (t & FpU<P>::MASK) : t;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) a.l[i] = borrow ? a.l[i] : d[i];
}
// Any value < 3q (nearly normalised limbs) -> the canonical saturated words of the same residue.
template <class P>
GMSM_HD Fp<P> fpu_canon_lt3q(FpU<P> a) {
    fpu_normalize(a);
    fpu_cond_sub_kq<P, 2>(a);
    fpu_cond_sub_kq<P, 1>(a);
    Fp<P> z;
    fpu_pack(a, z.l);
    return z;
}

// Exact test "value == 0 mod q" for a product-class value (normalised limbs, value < 3q): compares with 0, q, 2q.
template <class P>
GMSM_HD bool fpu_is_zero_lt3q(const FpU<P> &a) {
    uint32_t z0 = 0, z1 = 0, z2 = 0;
#pragma unroll
    for (int i = 0; i < P::UL; ++i) {
        z0 |= a.l[i];
        z1 |= a.l[i] ^ P::UQ1[i];
        z2 |= a.l[i] ^ P::UQ2[i];
    }
    return z0 == 0 || z1 == 0 || z2 == 0;
}

}  // namespace gmsm
