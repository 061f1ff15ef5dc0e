// Radix-2 butterflies of fr/fft on lazy limbs (host- and device-compilable; the kernels are in gmsm_fft.h).
//
// fr.Butterfly and the twiddle product (ecc/bn254/fr/fft/fft.go:198-330) on the saturated field cost a carry chain
// per partial product; on 29-bit limbs (gmsm_fieldu.h) the product is one v_mad_u64_u32 per partial product. Two
// things make the lazy form fit the FFT without conversions:
//   * DOMAIN. Vector elements stay what the reference holds - residues x*2^(32N) - merely re-cut into W-bit limbs on
//     load (fpu_unpack, no arithmetic). The lazy product divides by 2^(L*W), so the tables carry the difference:
//     every twiddle / coset factor is stored as the canonical element  2^(L*W-32N) * w^t  (BN254: 32 w^t). Then
//     fpu_mul(x*2^(32N), 2^(L*W-32N) w 2^(32N)) = (x w) * 2^(32N): still the reference's residue.
//   * CLASS. Every butterfly is Cooley-Tukey (product first): a value's bound grows by at most 2q per stage, a pass of up
//     to DIT_FREE_STAGES stages runs without any reduction, and an estimated quotient brings the values back below 2q
//     once per pass, on the store (round 4; rounds 2-3 ran the reference's Gentleman-Sande DIF with a top-limb
//     conditional subtraction per butterfly output, the class "A2").
#pragma once
#include "gmsm_fieldu.h"

namespace gmsm {

template <class P>
struct FftLz {
    using Fr = Fp<P>;
    using U = FpU<P>;
    static constexpr int L = P::UL, W = P::UW;
    static constexpr unsigned DOMAIN_SHIFT = (unsigned)(L * W - 32 * P::N);  // tables hold 2^DOMAIN_SHIFT * (factor)

    GMSM_HD static U load(const Fr &x) { return fpu_unpack<P>(x.l); }          // canonical residue -> limbs (< q)
    GMSM_HD static Fr store(const U &x) { return fpu_canon_lt3q<P>(x); }       // A2 -> canonical residue
    GMSM_HD static U mul(const U &x, const Fr &table_entry) { return fpu_mul(x, fpu_unpack<P>(table_entry.l)); }
    GMSM_HD static U mul(const U &x, const U &cut_entry) { return fpu_mul(x, cut_entry); }  // entry already cut into limbs

    // The Cooley-Tukey butterfly (ditFFT, fft.go:264-330: t = y w; (x, y) <- (x + t, x - t)) WITHOUT reductions. The
    // product comes first, so both outputs are "x plus something below 2q": t = y w < 1.2q whatever y is (y < 40q: 40/169
    // + 1), x + t < X + 1.2q and x - t + 2q < X + 2q. The bound of a value grows by at most 2q per stage instead of
    // doubling as in the Gentleman-Sande form (difFFT, fft.go:198-262), so a pass of up to DIT_FREE_STAGES stages needs
    // no conditional subtraction at all; the elements come back below 2q once per pass, on the store, by an estimated
    // quotient (reduce_big). Bounds: loaded value < 5.3q (anything that fits 2^(32N)); the stage without its product
    // (dit_one, the first of a transform): +8q; every other stage +2q: 5.3 + 8 + 2 * 10 < 34q.
    // LIMBS. The carry passes of the two outputs are a seventh of a butterfly's instructions, and only every SECOND stage
    // needs them: with carried inputs (limbs <= 2^W + 8) a stage leaves x + t <= 2 * 2^W + 8 and x + 2q' - t <= 4 * 2^W + 8
    // (the redundant 2q carries 2 * 2^W per limb besides its own < 2^W); the next stage can still multiply such a y - one
    // operand up to 4 * 2^W + 8 against an exactly normalised table entry keeps a column below L * 5 * 2^(2W) < 2^64 -
    // and its own sums stay below 8 * 2^W <= 2^32 until they are carried at its end. CARRY = false therefore alternates
    // with CARRY = true, the last stage of a pass being a CARRY = false one: the store's reduction (reduce_big, or the
    // final product) takes un-carried limbs.
    static_assert(W <= 29 && (unsigned long long)L * 5 < (1ull << (64 - 2 * W)), "column accumulator with an un-carried operand");
    static constexpr unsigned DIT_FREE_STAGES = 11;
    template <bool CARRY, class TW>
    GMSM_HD static void dit_free(U &x, U &y, const TW &w) {
        const U t = mul(y, w);  // < 1.2 q < 2q - 2 units: what the redundant 2q below admits; limbs exactly normalised
        U r, s;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            r.l[i] = x.l[i] + fpu_k2q<P>(i) - t.l[i];
            s.l[i] = x.l[i] + t.l[i];
        }
        if (CARRY) {
            fpu_carry(r);
            fpu_carry(s);
        }
        x = s;
        y = r;
    }
    // the stage of bit 0, whose twiddles are all one: (x, y) <- (x + y, x - y + 8q), y < 8q (the first stage of a DIT pass)
    GMSM_HD static void dit_one(U &x, U &y) {
        const U d = fpu_sub<P, 8>(x, y);
        x = fpu_add(x, y);
        y = d;
    }
    // v < 40q (limbs below 2^32, carried or not) -> the same residue below 2q + 2^-10 q, limbs fully normalised: k q is
    // taken off, k = floor(top(v) * floor(2^32 / (top(q) + 1)) / 2^32) <= floor(v / q), short of it by at most one (the
    // carries an un-carried value still owes its top limb are a few units of 2^(W(L-1)), far below q).
    GMSM_HD static U reduce_big(const U &v) {
        constexpr uint32_t QT1 = P::UQ1[L - 1] + 1u;
        constexpr uint32_t M = (uint32_t)(0x100000000ull / QT1);
        const uint32_t k = (uint32_t)(((uint64_t)v.l[L - 1] * M) >> 32);
        U r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int64_t t = (int64_t)v.l[i] - (int64_t)((uint64_t)k * P::UQ1[i]) + c;
            if (i < L - 1) {
                r.l[i] = (uint32_t)t & FpU<P>::MASK;
                c = t >> W;  // arithmetic shift: the borrow
            } else {
                r.l[i] = (uint32_t)t;
            }
        }
        return r;
    }
    // Stores of a pass that is not the transform's last: any representative that fits the 32N-bit element will do (the
    // next pass re-cuts it into limbs).
    GMSM_HD static Fr store_lazy_big(const U &a) {  // a < 40q (reduction-free passes)
        const U r = reduce_big(a);
        Fr z;
        fpu_pack(r, z.l);
        return z;
    }
    GMSM_HD static Fr store_big(const U &a) { return fpu_canon_lt3q<P>(reduce_big(a)); }  // canonical, from < 40q
};

}  // namespace gmsm
