// Radix-2 butterflies of fr/fft on lazy limbs (host- and device-compilable; the kernels are in gmsm_fft.h).
//
// fr.Butterfly and the twiddle product (ecc/bn254/fr/fft/fft.go:198-330) on the saturated field cost a carry chain
// per partial product; on 29-bit limbs (gmsm_fieldu.h) the product is one v_mad_u64_u32 per partial product. Two
// things make the lazy form fit the FFT without conversions:
//   * DOMAIN. Vector elements stay what the reference holds - residues x*2^(32N) - merely re-cut into W-bit limbs on
//     load (fpu_unpack, no arithmetic). The lazy product divides by 2^(L*W), so the tables carry the difference:
//     every twiddle / coset factor is stored as the canonical element  2^(L*W-32N) * w^t  (BN254: 32 w^t). Then
//     fpu_mul(x*2^(32N), 2^(L*W-32N) w 2^(32N)) = (x w) * 2^(32N): still the reference's residue.
//   * CLASS. Between products the values live in A2 = [0, 2q + D) (gmsm_fieldu.h: top-limb test, no normalisation);
//     a product of an A2 value (or of a difference < 6.5q) with a canonical table entry is < 1.1q. On store the value
//     (< 3q) is normalised exactly, so every output is the canonical element the reference computes.
// D after s consecutive additions is <= 2^s * 6 * 2^(W(L-1)); FFT_MAX_CHAIN stages per pass keep it below q/8 for the
// 254/255-bit fields; BW6-761's scalar field (q's top limb has 13 bits) runs the TIGHT form, D <= 6 * 2^(W(L-1)).
#pragma once
#include "gmsm_fieldu.h"

namespace gmsm {

constexpr unsigned FFT_MAX_CHAIN = 11;  // most stages k_fft_pass_lz runs between a load and a store

template <class P>
struct FftLz {
    using Fr = Fp<P>;
    using U = FpU<P>;
    static constexpr int L = P::UL, W = P::UW;
    // top limb of q, floor(q / 2^(W(L-1))): the chain excess D = 2^FFT_MAX_CHAIN * 6 units must stay below an eighth of it
    static constexpr bool TIGHT = ((uint64_t)6 << FFT_MAX_CHAIN) * 8 > (uint64_t)P::UQ[L - 1];
    static constexpr unsigned DOMAIN_SHIFT = (unsigned)(L * W - 32 * P::N);  // tables hold 2^DOMAIN_SHIFT * (factor)

    GMSM_HD static U load(const Fr &x) { return fpu_unpack<P>(x.l); }          // canonical residue -> limbs (< q)
    GMSM_HD static Fr store(const U &x) { return fpu_canon_lt3q<P>(x); }       // A2 -> canonical residue
    GMSM_HD static U mul(const U &x, const Fr &table_entry) { return fpu_mul(x, fpu_unpack<P>(table_entry.l)); }
    GMSM_HD static U mul(const U &x, const U &cut_entry) { return fpu_mul(x, cut_entry); }  // entry already cut into limbs

    // decimation in frequency (difFFT, fft.go:198-262): (x, y) <- (x + y, (x - y) w)
    template <class TW>  // TW = Fr (packed table entry, re-cut here) or U (pre-cut table entry)
    GMSM_HD static void dif(U &x, U &y, const TW &w) {
        const U d = fpu_sub<P, 4>(x, y);  // x - y + 4q < 6q + D, y < 4q
        x = fpu_add_a2<P, TIGHT>(x, y);
        y = mul(d, w);                    // < 1.1 q
    }
    // decimation in time (ditFFT, fft.go:264-330): t = y w; (x, y) <- (x + t, x - t)
    template <class TW>
    GMSM_HD static void dit(U &x, U &y, const TW &w) {
        const U t = mul(y, w);  // < 1.1 q
        y = fpu_sub_a2<P, TIGHT>(x, t);
        x = fpu_add_a2<P, TIGHT>(x, t);
    }

    // ---- the same butterfly WITHOUT reductions (round 4). In the Cooley-Tukey form the product comes first, so both
    // outputs are "x plus something below 2q": t = y w < 1.2q whatever y is (y < 32q: 32/169 + 1), x + t < X + 1.2q and
    // x - t + 2q < X + 2q. The bound of a value grows by at most 2q per stage instead of doubling as in the
    // Gentleman-Sande form, so a pass of up to DIT_FREE_STAGES stages needs no conditional subtraction at all (two
    // top-limb tests with their masked additions per butterfly gone); the elements come back below 2q once per pass, on
    // the store, by an estimated quotient (reduce_big). Bounds: loaded value < 5.3q (anything that fits 2^(32N)); bit-0
    // stage without its product (dit_one): +8q; every other stage +2q: 5.3 + 8 + 2 * 10 < 34q < 2^(32N+3).
    static constexpr unsigned DIT_FREE_STAGES = 11;
    template <class TW>
    GMSM_HD static void dit_free(U &x, U &y, const TW &w) {
        const U t = mul(y, w);  // < 1.2 q < 2q - 2 units: what the redundant 2q below admits
        U r;
#pragma unroll
        for (int i = 0; i < L; ++i) r.l[i] = x.l[i] + fpu_k2q<P>(i) - t.l[i];
        fpu_carry(r);
        x = fpu_add(x, t);
        y = r;
    }
    // the stage of bit 0, whose twiddles are all one: (x, y) <- (x + y, x - y + 8q), y < 8q (the first stage of a DIT pass)
    GMSM_HD static void dit_one(U &x, U &y) {
        const U d = fpu_sub<P, 8>(x, y);
        x = fpu_add(x, y);
        y = d;
    }
    // v < 40q (nearly normalised limbs) -> the same residue below 2q + 2^-10 q, limbs fully normalised: k q is taken
    // off, k = floor(top(v) * floor(2^32 / (top(q) + 1)) / 2^32) <= floor(v / q), short of it by at most one.
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
    // next pass re-cuts it into limbs). a: class A2 (Gentleman-Sande passes) -> exactly below 2q.
    GMSM_HD static Fr store_lazy_a2(U a) {
        fpu_normalize(a);
        fpu_cond_sub_kq<P, 2>(a);
        Fr z;
        fpu_pack(a, z.l);
        return z;
    }
    GMSM_HD static Fr store_lazy_big(const U &a) {  // a < 40q (reduction-free passes)
        const U r = reduce_big(a);
        Fr z;
        fpu_pack(r, z.l);
        return z;
    }
    GMSM_HD static Fr store_big(const U &a) { return fpu_canon_lt3q<P>(reduce_big(a)); }  // canonical, from < 40q
};

}  // namespace gmsm
