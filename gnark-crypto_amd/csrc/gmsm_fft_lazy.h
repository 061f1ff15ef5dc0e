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

    // decimation in frequency (difFFT, fft.go:198-262): (x, y) <- (x + y, (x - y) w)
    GMSM_HD static void dif(U &x, U &y, const Fr &w) {
        const U d = fpu_sub<P, 4>(x, y);  // x - y + 4q < 6q + D, y < 4q
        x = fpu_add_a2<P, TIGHT>(x, y);
        y = mul(d, w);                    // < 1.1 q
    }
    // decimation in time (ditFFT, fft.go:264-330): t = y w; (x, y) <- (x + t, x - t)
    GMSM_HD static void dit(U &x, U &y, const Fr &w) {
        const U t = mul(y, w);  // < 1.1 q
        y = fpu_sub_a2<P, TIGHT>(x, t);
        x = fpu_add_a2<P, TIGHT>(x, t);
    }
};

}  // namespace gmsm
