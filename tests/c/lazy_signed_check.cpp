// Host-side check of the signed-limb mixed addition (gmsm_curveu.h, madd_s: differences without K*q offsets or carry
// passes, product scans on signed 64-bit columns) against the unsigned bound-tracked form it replaces in the accumulation
// loop (madd_u), on the three base fields in scope (9 x 29, 14 x 28 and 28 x 28 bit limbs). The formulas are algebraic
// identities, so arbitrary field elements serve as coordinates: chains of additions from infinity with random and
// edge-valued (0, 1, q - 1) coordinates, both signs, the same point twice in a row (the doubling branch, which answers in
// the unsigned class and is continued on signed limbs) and a point followed by its negation (back to infinity). After
// every step
//   * both accumulators hold the same residues (the signed one after lz_acc_finish, compared in canonical form);
//   * the signed accumulator's limbs stay within what the signed product scans admit, +-(2^W + 2^(32-W));
//   * the finished record is in the class the records in memory use: non-negative nearly normalised limbs,
//     x, y < 11q, zz < 3q, zzz < 6q.
// The Fp2 form (madd_ts, BN254 and BLS12-381 G2) is checked the same way against the exact reduced-class addition madd_g;
// its finished records must be exactly normalised and below 4q, and stay so through another lz_rec_fresh.
// Exit code = number of mismatches (capped).
// Build: clang++ -O2 -std=c++17 -D__host__= -D__device__= -D__noinline__= -D__forceinline__=inline tests/c/lazy_signed_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../gnark-crypto_amd/csrc/gmsm_params32.h"
#include "../../gnark-crypto_amd/csrc/gmsm_curveu.h"
using namespace gmsm;

template <class P>
static FpU<P> pick(std::mt19937_64 &g) {
    Fp<P> x;
    const unsigned kind = (unsigned)(g() % 16);
    for (int i = 0; i < P::N; ++i) x.l[i] = kind == 0 ? 0u : kind == 1 ? P::Q[i] : (uint32_t)g();
    if (kind == 1) x.l[0] -= 1;                                   // q - 1
    else if (kind == 2) { memset(&x, 0, sizeof x); x.l[0] = 1; }  // 1
    else if (kind != 0) x.l[P::N - 1] %= P::Q[P::N - 1];          // below q
    return fpu_unpack<P>(x.l);
}

template <class P>
static bool same(const FpU<P> &a, const FpU<P> &b) {
    const Fp<P> x = fpu_to_sat<P, true>(a), y = fpu_to_sat<P, true>(b);
    return memcmp(&x, &y, sizeof x) == 0;
}
template <class P>
static bool signed_limbs_ok(const FpU<P> &a) {
    const int64_t lim = ((int64_t)1 << P::UW) + ((int64_t)1 << (32 - P::UW));
    for (int i = 0; i < P::UL - 1; ++i) {
        const int64_t v = (int32_t)a.l[i];
        if (v > lim || v < -lim) return false;
    }
    return true;
}
// non-negative, nearly normalised, value < K q
template <class P>
static bool in_class(const FpU<P> &a, unsigned K) {
    for (int i = 0; i < P::UL - 1; ++i)
        if (a.l[i] > (1u << P::UW) + (1u << (32 - P::UW))) return false;
    FpU<P> n = a;
    fpu_normalize(n);
    if ((int32_t)n.l[P::UL - 1] < 0) return false;
    return (uint64_t)n.l[P::UL - 1] < (uint64_t)K * ((uint64_t)P::UQ1[P::UL - 1] + 1ull) - K;  // top limb alone decides, with slack
}

template <class P>
static int check(const char *name, int chains) {
    using U = FpU<P>;
    static_assert(LzSigned<U>::value, "the prime-field groups run madd_s");
    std::mt19937_64 g(0x51 + P::UL);
    int bad = 0, steps = 0, doubled = 0, cancelled = 0;
    for (int chain = 0; chain < chains; ++chain) {
        XYZZL<U> t, r;
        bool tinf = true, rinf = true;
        U px = pick<P>(g), py = pick<P>(g);
        bool neg = false;
        const int len = 2 + (int)(g() % 40);
        for (int k = 0; k < len; ++k) {
            const unsigned what = (unsigned)(g() % 12);
            if (what == 0 && k > 0) {            // the same affine point again: right after the first addition that doubles
                if (k == 1) ++doubled;
            } else if (what == 1 && k > 0) {     // its negation
                neg = !neg;
            } else {
                px = pick<P>(g);
                py = pick<P>(g);
                neg = (g() & 1) != 0;
            }
            const bool was_inf = rinf;
            lz_madd_acc<true>(t, tinf, px, py, neg);   // madd_s
            madd_u<P, true>(r, rinf, px, py, neg);
            ++steps;
            bool ok = tinf == rinf;
            if (ok && !tinf) {
                ok = signed_limbs_ok(t.x) && signed_limbs_ok(t.y) && signed_limbs_ok(t.zz) && signed_limbs_ok(t.zzz);
                XYZZL<U> f = t;
                lz_acc_finish(f, false);
                ok = ok && same(f.x, r.x) && same(f.y, r.y) && same(f.zz, r.zz) && same(f.zzz, r.zzz);
                ok = ok && in_class(f.x, 11) && in_class(f.y, 11) && in_class(f.zz, 3) && in_class(f.zzz, 6);
                // the bucket path: the signed accumulator is stored as it is and its readers run lz_rec_fresh - once, or
                // again on a record that already went through it (a bucket the fix-up or a range merge re-stored)
                XYZZL<U> rec = t;
                lz_rec_fresh(rec);
                ok = ok && memcmp(&rec, &f, sizeof rec) == 0;
                lz_rec_fresh(rec);
                lz_rec_fresh(rec);
                ok = ok && same(rec.x, r.x) && same(rec.y, r.y) && same(rec.zzz, r.zzz) && in_class(rec.x, 19) && in_class(rec.y, 19) &&
                     in_class(rec.zzz, 14);
                // mimic the fixed-base walk now and then: nothing is flushed for the whole chain; and the unsigned class
                // is a subset of the signed one, so continuing from the finished record must work too
                if ((g() & 15) == 0) t = f;
            }
            if (!was_inf && rinf) ++cancelled;
            if (!ok && bad++ < 5) printf("%s: chain %d step %d: mismatch (inf %d/%d)\n", name, chain, k, (int)tinf, (int)rinf);
        }
    }
    printf("%s: %d additions, %d doublings, %d cancellations, %d mismatches\n", name, steps, doubled, cancelled, bad);
    return bad;
}

// ---- Fp2: madd_ts against the exact reduced-class form madd_g (BN254, BLS12-381) ----
template <class P>
static bool in_r(const FpU<P> &a) {  // exactly normalised, < 4q
    for (int i = 0; i < P::UL - 1; ++i)
        if (a.l[i] >> P::UW) return false;
    for (int i = P::UL - 1; i >= 0; --i) {
        if (a.l[i] < P::UQ4[i]) return true;
        if (a.l[i] > P::UQ4[i]) return false;
    }
    return false;
}
template <class P>
static bool same2(const Fp2U<P> &a, const Fp2U<P> &b) { return same(a.a0, b.a0) && same(a.a1, b.a1); }
template <class P>
static bool signed2_ok(const Fp2U<P> &a) { return signed_limbs_ok(a.a0) && signed_limbs_ok(a.a1); }
template <class P>
static bool in_r2(const Fp2U<P> &a) { return in_r(a.a0) && in_r(a.a1); }

template <class P>
static int check2(const char *name, int chains) {
    using U = Fp2U<P>;
    static_assert(LzSigned<U>::value, "the Fp2 groups run madd_ts");
    std::mt19937_64 g(0x52 + P::UL);
    int bad = 0, steps = 0, doubled = 0, cancelled = 0;
    for (int chain = 0; chain < chains; ++chain) {
        XYZZL<U> t, r;
        bool tinf = true, rinf = true;
        U px{pick<P>(g), pick<P>(g)}, py{pick<P>(g), pick<P>(g)};
        bool neg = false;
        const int len = 2 + (int)(g() % 40);
        for (int k = 0; k < len; ++k) {
            const unsigned what = (unsigned)(g() % 12);
            if (what == 0 && k > 0) {
                if (k == 1) ++doubled;
            } else if (what == 1 && k > 0) {
                neg = !neg;
            } else {
                px = U{pick<P>(g), pick<P>(g)};
                py = U{pick<P>(g), pick<P>(g)};
                neg = (g() & 1) != 0;
            }
            const bool was_inf = rinf;
            lz_madd_acc<true>(t, tinf, px, py, neg);   // madd_ts
            madd_g<U, true>(r, rinf, px, py, neg);
            ++steps;
            bool ok = tinf == rinf;
            if (ok && !tinf) {
                ok = signed2_ok(t.x) && signed2_ok(t.y) && signed2_ok(t.zz) && signed2_ok(t.zzz);
                XYZZL<U> f = t;
                lz_acc_finish(f, false);
                ok = ok && same2(f.x, r.x) && same2(f.y, r.y) && same2(f.zz, r.zz) && same2(f.zzz, r.zzz);
                ok = ok && in_r2(f.x) && in_r2(f.y) && in_r2(f.zz) && in_r2(f.zzz);
                XYZZL<U> rec = f;   // a record that already is in R stays in R, same residues
                lz_rec_fresh(rec);
                ok = ok && same2(rec.x, r.x) && same2(rec.y, r.y) && same2(rec.zz, r.zz) && same2(rec.zzz, r.zzz) && in_r2(rec.x) &&
                     in_r2(rec.y) && in_r2(rec.zz) && in_r2(rec.zzz);
                if ((g() & 15) == 0) t = f;
            }
            if (!was_inf && rinf) ++cancelled;
            if (!ok && bad++ < 5) printf("%s: chain %d step %d: mismatch (inf %d/%d)\n", name, chain, k, (int)tinf, (int)rinf);
        }
    }
    printf("%s: %d additions, %d doublings, %d cancellations, %d mismatches\n", name, steps, doubled, cancelled, bad);
    return bad;
}

int main() {
    int bad = check<bn254_fp_params>("bn254 fp", 40000);
    bad += check<bls12_381_fp_params>("bls12-381 fp", 20000);
    bad += check<bw6_761_fp_params>("bw6-761 fp", 6000);
    bad += check2<bn254_fp_params>("bn254 fp2", 20000);
    bad += check2<bls12_381_fp_params>("bls12-381 fp2", 10000);
    return bad > 100 ? 100 : bad;
}
