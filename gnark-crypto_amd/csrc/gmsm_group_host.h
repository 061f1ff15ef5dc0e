// Host-side arithmetic of one (curve, group): the Horner fold of the window totals (msmReduceChunk), the merge of
// point-sharded total sets, on-curve input generation, the fixed-base table and the powers of Fold - plain C++ over the
// saturated field of gmsm_field.h, no device code. struct Group (gmsm_group.h) derives from it.
#pragma once
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "gmsm_context.h"
#include "gmsm_curve.h"

namespace gmsm {

template <class F_, class FrP_>
struct GroupHost {
    using F = F_;
    using FrP = FrP_;
    using Aff = Affine<F>;
    using Ext = XYZZ<F>;
    using J = Jac<F>;
    static constexpr unsigned FR_BITS = FrP::BITS;

    // Point-sharded MultiExp: `nsets` ranks each hold the nwin window totals of their own slice of the points; window w
    // of the whole MultiExp is the sum over the ranks (g1JacExtended.add, g1.go:736), then the usual fold.
    static J fold_sets(const Ext *sets, unsigned nsets, unsigned c) {
        const unsigned nwin = num_windows(FR_BITS, c);
        std::vector<Ext> totals(sets, sets + nwin);
        for (unsigned s = 1; s < nsets; ++s)
            for (unsigned w = 0; w < nwin; ++w) xyzz_add(totals[w], sets[(size_t)s * nwin + w]);
        return fold(totals.data(), c);
    }

    // msmReduceChunk (multiexp.go:302-315): Horner from the top window down. The (nwin - 1) c doublings are a serial chain
    // on the host after the device has finished - 83 us of a 1.98 ms BN254 G1 call, 0.28 ms of a BN254 G2 call - so they run
    // in Jacobian coordinates (2M + 5S per doubling against the 6M + 3S of the extended form; over Fp2 a square is two
    // base products, a product three: 16 against 24) and the running sum changes form around each window's addition.
    // Windows at infinity (all but the first under window tables) cost nothing.
    static J fold(const Ext *totals, unsigned c, unsigned nwin = 0 /* 0: the windows of a full scalar */) {
        if (nwin == 0) nwin = num_windows(FR_BITS, c);
        J acc = jac_from_xyzz(totals[nwin - 1]);
        for (int j = (int)nwin - 2; j >= 0; --j) {
            if (!acc.z.is_zero())
                for (unsigned l = 0; l < c; ++l) acc = jac_double(acc);
            if (totals[j].zz.is_zero()) continue;
            Ext e = acc.z.is_zero() ? Ext::infinity() : xyzz_from_jac(acc);
            xyzz_add(e, totals[j]);
            acc = jac_from_xyzz(e);
        }
        if (acc.z.is_zero()) acc = J{F::one(), F::one(), F::zero()};
        return acc;
    }

    // out[i] = [k0 + i*k1] base for i < n (affine). Host-side, multi-threaded: start point by double-and-add, then
    // repeated mixed addition of step = [k1]base with block-wise batch normalisation (one inversion per block).
    // Utility for building SRS-like on-curve bases (cf. BatchScalarMultiplicationG1, ecc/bn254/g1.go:1039, and the
    // i*G walk of multiexp_test.go:40-46); bench.py uses it for its synthetic inputs.
    static Ext scalar_mul(const Aff &a, const uint64_t *k, int klimbs) {
        Ext acc = Ext::infinity();
        for (int i = klimbs * 64 - 1; i >= 0; --i) {
            acc = xyzz_double(acc);
            if ((k[i / 64] >> (i % 64)) & 1) xyzz_add_mixed(acc, a, false);
        }
        return acc;
    }
    static void batch_to_affine(Aff *out, const Ext *in, size_t count, F *scratch) {
        F acc = F::one();
        for (size_t i = 0; i < count; ++i) {
            scratch[i] = acc;
            if (!in[i].zzz.is_zero()) acc = fp_mul(acc, in[i].zzz);
        }
        F inv = fp_inv(acc);
        for (size_t i = count; i-- > 0;) {
            if (in[i].zzz.is_zero()) {
                out[i] = Aff{F::zero(), F::zero()};
                continue;
            }
            F zi = fp_mul(inv, scratch[i]);  // 1/zzz_i
            inv = fp_mul(inv, in[i].zzz);
            F izz = fp_mul(fp_mul(fp_sqr(zi), in[i].zz), in[i].zz);  // zz^2/zzz^2 = 1/zz
            out[i].x = fp_mul(in[i].x, izz);
            out[i].y = fp_mul(in[i].y, zi);
        }
    }
    static void generate_points(const uint64_t *base_limbs, const uint64_t *k0, const uint64_t *k1, int klimbs, size_t n,
                                int nthreads, uint64_t *out_limbs) {
        Aff base;
        memcpy(&base, base_limbs, sizeof base);
        Aff *out = reinterpret_cast<Aff *>(out_limbs);
        Aff step;
        {
            Ext s = scalar_mul(base, k1, klimbs);
            F scratch;
            batch_to_affine(&step, &s, 1, &scratch);
        }
        if (nthreads < 1) nthreads = 1;
        const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) {
            const size_t s = (size_t)t * per, e = std::min(n, s + per);
            if (s >= e) break;
            th.emplace_back([=, &base, &step]() {
                constexpr size_t BLK = 1024;
                std::vector<Ext> blk(BLK);
                std::vector<F> scr(BLK);
                Ext cur = scalar_mul(base, k0, klimbs);
                uint64_t s64 = (uint64_t)s;
                Ext off = scalar_mul(step, &s64, 1);
                xyzz_add(cur, off);
                for (size_t i = s; i < e;) {
                    const size_t cnt = std::min(BLK, e - i);
                    for (size_t k = 0; k < cnt; ++k) {
                        blk[k] = cur;
                        xyzz_add_mixed(cur, step, false);
                    }
                    batch_to_affine(out + i, blk.data(), cnt, scr.data());
                    i += cnt;
                }
            });
        }
        for (auto &t : th) t.join();
    }

    // table[j][d-1] = d * 2^(c*j) * base for every window j and d = 1..nb (host, threads over windows), Go layout.
    static void build_fixed_base_table(const Aff &base, uint32_t c, uint32_t nwin, uint32_t nb, int nthreads, std::vector<Aff> &table) {
        table.resize((size_t)nwin * nb);
        std::vector<Aff> win_base(nwin);  // 2^(c*j) * base
        {
            std::vector<Ext> wb(nwin);
            Ext cur = Ext::infinity();
            xyzz_add_mixed(cur, base, false);
            for (uint32_t j = 0; j < nwin; ++j) {
                wb[j] = cur;
                for (uint32_t l = 0; l < c; ++l) cur = xyzz_double(cur);
            }
            std::vector<F> scr(nwin);
            batch_to_affine(win_base.data(), wb.data(), nwin, scr.data());
        }
        if (nthreads < 1) nthreads = 1;
        if ((uint32_t)nthreads > nwin) nthreads = (int)nwin;
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t] {
                std::vector<Ext> row(nb);
                std::vector<F> scr(nb);
                for (uint32_t j = (uint32_t)t; j < nwin; j += (uint32_t)nthreads) {
                    Ext cur = Ext::infinity();
                    for (uint32_t d = 0; d < nb; ++d) {
                        xyzz_add_mixed(cur, win_base[j], false);
                        row[d] = cur;
                    }
                    batch_to_affine(table.data() + (size_t)j * nb, row.data(), nb, scr.data());
                }
            });
        for (auto &x : th) x.join();
    }

    // (*G1Jac).Fold (multiexp.go:331-340): the scalars 1, g, g^2, ... (Montgomery fr products on the host); the engine then
    // runs the MultiExp entry over them.
    static void fold_powers(const uint64_t *coeff, size_t n, uint64_t *out_scalars) {
        using Fr = Fp<FrP>;
        Fr *scalars = reinterpret_cast<Fr *>(out_scalars);
        Fr g, s = Fr::one();
        memcpy(&g, coeff, sizeof g);
        for (size_t i = 0; i < n; ++i) {
            scalars[i] = s;
            s = fp_mul(s, g);
        }
    }
};

}  // namespace gmsm
