// Host-side check of the bound-tracked Fp2 mixed addition (gmsm_curveu.h, madd_t: limb-parallel subtractions, Y3 as one
// four-product form per component) against the reduced-class form it replaces in the accumulation loop (madd_g: exact
// normalised arithmetic in [0, 4q)), on BLS12-381's base field. The formulas are algebraic identities, so arbitrary
// field elements serve as coordinates: chains of additions from infinity with random and edge-valued (0, 1, q - 1)
// coordinates, both signs, the same point twice in a row (the doubling branch) and a point followed by its negation
// (back to infinity); after every step both accumulators must hold the same residues, and the tracked one - after
// lz_acc_finish - exactly normalised limbs below 4q. Exit code = number of mismatches (capped).
// (Since round 4 the shipped accumulation loop runs madd_ts on signed limbs - tests/c/lazy_signed_check.cpp; madd_t is
// what -DGMSM_SIGNED_MADD2=0 builds, and this check is compiled that way.)
// Build: clang++ -O2 -std=c++17 -DGMSM_SIGNED_MADD2=0 -D__host__= -D__device__= -D__noinline__= -D__forceinline__=inline tests/c/lazy_g2_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../gnark-crypto_amd/csrc/gmsm_params32.h"
#include "../../gnark-crypto_amd/csrc/gmsm_curveu.h"
using namespace gmsm;

using P = bls12_381_fp_params;
using U = Fp2U<P>;

static FpU<P> pick(std::mt19937_64 &g) {
    Fp<P> x;
    const unsigned kind = (unsigned)(g() % 16);
    for (int i = 0; i < P::N; ++i) x.l[i] = kind == 0 ? 0u : kind == 1 ? P::Q[i] : (uint32_t)g();
    if (kind == 1) x.l[0] -= 1;                                   // q - 1
    else if (kind == 2) { memset(&x, 0, sizeof x); x.l[0] = 1; }  // 1
    else if (kind != 0) x.l[P::N - 1] %= P::Q[P::N - 1];          // below q
    return fpu_unpack<P>(x.l);
}

static bool same(const FpU<P> &a, const FpU<P> &b) {
    const Fp<P> x = fpu_to_sat<P, true>(a), y = fpu_to_sat<P, true>(b);
    return memcmp(&x, &y, sizeof x) == 0;
}
static bool same2(const U &a, const U &b) { return same(a.a0, b.a0) && same(a.a1, b.a1); }
static bool in_r(const FpU<P> &a) {  // exactly normalised, < 4q
    for (int i = 0; i < P::UL - 1; ++i)
        if (a.l[i] >> P::UW) return false;
    for (int i = P::UL - 1; i >= 0; --i) {
        if (a.l[i] < P::UQ4[i]) return true;
        if (a.l[i] > P::UQ4[i]) return false;
    }
    return false;
}

int main() {
    static_assert(LzTracked<U>::value, "BLS12-381 G2 runs the tracked form");
    static_assert(!LzTracked<Fp2U<bn254_fp_params>>::value, "BN254 (7 spare bits) stays on the reduced class");
    std::mt19937_64 g(0x62);
    int bad = 0, steps = 0, doubled = 0, cancelled = 0;
    for (int chain = 0; chain < 3000; ++chain) {
        XYZZL<U> t, r;
        bool tinf = true, rinf = true;
        U px{pick(g), pick(g)}, py{pick(g), pick(g)};
        bool neg = false;
        const int len = 2 + (int)(g() % 40);
        for (int k = 0; k < len; ++k) {
            const unsigned what = (unsigned)(g() % 12);
            if (what == 0 && k > 0) {            // the same affine point again: right after the first addition that doubles
                if (k == 1) ++doubled;
            } else if (what == 1 && k > 0) {     // its negation
                neg = !neg;
            } else {
                px = U{pick(g), pick(g)};
                py = U{pick(g), pick(g)};
                neg = (g() & 1) != 0;
            }
            const bool was_inf = rinf;
            madd_t<P, true>(t, tinf, px, py, neg);
            madd_g<U, true>(r, rinf, px, py, neg);
            ++steps;
            bool ok = tinf == rinf;
            if (ok && !tinf) {
                XYZZL<U> f = t;
                lz_acc_finish(f, false);
                ok = same2(f.x, r.x) && same2(f.y, r.y) && same2(f.zz, r.zz) && same2(f.zzz, r.zzz);
                ok = ok && in_r(f.x.a0) && in_r(f.x.a1) && in_r(f.y.a0) && in_r(f.y.a1) && in_r(f.zz.a0) && in_r(f.zz.a1) &&
                     in_r(f.zzz.a0) && in_r(f.zzz.a1);
                // mimic the kernel now and then: the record is flushed and the walk continues from the reduced class
                if ((g() & 7) == 0) t = f;
            }
            if (!was_inf && rinf) ++cancelled;
            if (!ok && bad++ < 5) printf("chain %d step %d: mismatch (inf %d/%d)\n", chain, k, (int)tinf, (int)rinf);
        }
    }
    printf("bls12-381 fp2: %d additions, %d doublings, %d cancellations, %d mismatches\n", steps, doubled, cancelled, bad);
    return bad > 100 ? 100 : bad;
}
