// Host-side check of the lazy-limb FFT butterflies (gmsm_fft_lazy.h) against the saturated canonical field
// (gmsm_field.h = what the reference computes): full radix-2 transforms of FFT_MAX_CHAIN stages without any reduction in
// between (the longest chain a device pass runs), DIF and DIT, on random inputs and on inputs that maximise the growth
// of the lazy class (all q - 1, all zero, alternating). Every intermediate is also checked to stay inside the class the
// header promises: nearly normalised limbs, value < 3q. Exit code = number of mismatches (capped).
// Build: g++ -O2 -std=c++17 -D__host__= -D__device__= -D__noinline__= tests/c/lazy_fft_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../gnark-crypto_amd/csrc/gmsm_params32.h"
#include "../../gnark-crypto_amd/csrc/gmsm_fft_lazy.h"
using namespace gmsm;

template <class P>
static bool in_class(const FpU<P> &a) {
    for (int i = 0; i < P::UL - 1; ++i)
        if (a.l[i] > (1u << P::UW) + (1u << (32 - P::UW))) return false;
    FpU<P> n = a;
    fpu_normalize(n);
    for (int i = P::UL - 1; i >= 0; --i) {  // n < 3q ?
        if (n.l[i] < P::UQ3[i]) return true;
        if (n.l[i] > P::UQ3[i]) return false;
    }
    return false;
}

template <class P>
static int check(const char *name) {
    using Fr = Fp<P>;
    using Z = FftLz<P>;
    constexpr unsigned LOG = FFT_MAX_CHAIN;
    constexpr size_t N = (size_t)1 << LOG;
    std::mt19937_64 g(0xf17);
    Fr root;
    for (int i = 0; i < P::N; ++i) root.l[i] = P::ROOT_OF_UNITY[i];
    for (unsigned i = 0; i < P::MAX_ORDER - LOG; ++i) root = fp_sqr(root);
    Fr shift = Fr::one();
    for (unsigned i = 0; i < Z::DOMAIN_SHIFT; ++i) shift = fp_dbl(shift);
    std::vector<Fr> tw(N / 2), twz(N / 2);
    tw[0] = Fr::one();
    for (size_t t = 1; t < N / 2; ++t) tw[t] = fp_mul(tw[t - 1], root);
    for (size_t t = 0; t < N / 2; ++t) twz[t] = fp_mul(tw[t], shift);
    Fr qm1;  // q - 1
    for (int i = 0; i < P::N; ++i) qm1.l[i] = P::Q[i];
    qm1.l[0] -= 1;
    int bad = 0, cls = 0;
    for (int pattern = 0; pattern < 6; ++pattern) {
        std::vector<Fr> a(N);
        for (size_t i = 0; i < N; ++i) {
            if (pattern == 0) a[i] = qm1;
            else if (pattern == 1) a[i] = Fr::zero();
            else if (pattern == 2) a[i] = (i & 1) ? qm1 : Fr::zero();
            else {
                for (int k = 0; k < P::N; ++k) a[i].l[k] = (uint32_t)g();
                a[i].l[P::N - 1] %= P::Q[P::N - 1];
            }
        }
        for (int dif = 0; dif < 2; ++dif) {
            std::vector<Fr> s = a;
            std::vector<FpU<P>> z(N);
            for (size_t i = 0; i < N; ++i) z[i] = Z::load(a[i]);
            for (unsigned st = 0; st < LOG; ++st) {
                const unsigned b = dif ? LOG - 1 - st : st;  // bit paired by this stage
                for (size_t q = 0; q < N / 2; ++q) {
                    const size_t i0 = ((q >> b) << (b + 1)) | (q & (((size_t)1 << b) - 1)), i1 = i0 | ((size_t)1 << b);
                    const size_t j = i0 & (((size_t)1 << b) - 1), t = j << (LOG - 1 - b);
                    if (dif) {
                        const Fr d = fp_mul(fp_sub(s[i0], s[i1]), tw[t]);
                        s[i0] = fp_add(s[i0], s[i1]);
                        s[i1] = d;
                        Z::dif(z[i0], z[i1], twz[t]);
                    } else {
                        const Fr tt = fp_mul(s[i1], tw[t]);
                        s[i1] = fp_sub(s[i0], tt);
                        s[i0] = fp_add(s[i0], tt);
                        Z::dit(z[i0], z[i1], twz[t]);
                    }
                    if (!in_class<P>(z[i0]) || !in_class<P>(z[i1])) ++cls;
                }
            }
            for (size_t i = 0; i < N; ++i) {
                const Fr o = Z::store(z[i]);
                if (memcmp(&o, &s[i], sizeof o) != 0 && bad++ < 5) printf("%s: pattern %d dif %d: element %zu differs\n", name, pattern, dif, i);
            }
        }
        // Round 4: the reduction-free DIT pass (dit_one for bit 0, dit_free above it) started from the WORST input the
        // device ever loads - a lazy representative just below 2^(32N), i.e. the canonical value plus as many q as fit -
        // and ended by store_big / store_lazy_big; the value must stay below 40q all the way (limbs nearly normalised).
        {
            std::vector<Fr> s = a;
            std::vector<FpU<P>> z(N);
            for (size_t i = 0; i < N; ++i) {
                z[i] = Z::load(a[i]);
                for (int rep = 0; rep < 8; ++rep) {  // + q while the sum still fits 32N bits
                    FpU<P> t;
                    for (int k = 0; k < P::UL; ++k) t.l[k] = z[i].l[k] + P::UQ1[k];
                    fpu_normalize(t);
                    if ((uint64_t)t.l[P::UL - 1] >> (32 * P::N - P::UW * (P::UL - 1)) != 0) break;
                    z[i] = t;
                }
            }
            for (unsigned st = 0; st < LOG; ++st) {
                const unsigned b = st;
                for (size_t q = 0; q < N / 2; ++q) {
                    const size_t i0 = ((q >> b) << (b + 1)) | (q & (((size_t)1 << b) - 1)), i1 = i0 | ((size_t)1 << b);
                    const size_t j = i0 & (((size_t)1 << b) - 1), t = j << (LOG - 1 - b);
                    const Fr tt = fp_mul(s[i1], tw[t]);
                    s[i1] = fp_sub(s[i0], tt);
                    s[i0] = fp_add(s[i0], tt);
                    if (b == 0) Z::dit_one(z[i0], z[i1]);
                    else Z::dit_free(z[i0], z[i1], twz[t]);
                    for (const FpU<P> *v : {&z[i0], &z[i1]}) {
                        for (int k = 0; k < P::UL - 1; ++k)
                            if (v->l[k] > (1u << P::UW) + (1u << (32 - P::UW))) ++cls;
                        // below 40q: top limb below 40 * (top(q) + 1)
                        if ((uint64_t)v->l[P::UL - 1] >= 40ull * (P::UQ1[P::UL - 1] + 1ull)) ++cls;
                    }
                }
            }
            for (size_t i = 0; i < N; ++i) {
                const Fr o = Z::store_big(z[i]);
                if (memcmp(&o, &s[i], sizeof o) != 0 && bad++ < 5) printf("%s: pattern %d free DIT: element %zu differs\n", name, pattern, i);
                // the lazy store followed by a canonicalising reload must name the same element, and fit below 2q + q/512
                const Fr lz = Z::store_lazy_big(z[i]);
                const Fr back = Z::store(Z::load(lz));
                if (memcmp(&back, &s[i], sizeof back) != 0 && bad++ < 5) printf("%s: pattern %d lazy store: element %zu differs\n", name, pattern, i);
                FpU<P> a2 = Z::load(o);  // class A2 lazy store of a canonical value plus q: below 2q exactly
                for (int k = 0; k < P::UL; ++k) a2.l[k] += P::UQ1[k];
                const Fr la = Z::store_lazy_a2(a2);
                const Fr back2 = Z::store(Z::load(la));
                if (memcmp(&back2, &s[i], sizeof back2) != 0 && bad++ < 5) printf("%s: pattern %d a2 lazy store: element %zu differs\n", name, pattern, i);
            }
        }
    }
    printf("%s: tight=%d shift=%u: %d mismatches, %d class violations\n", name, (int)Z::TIGHT, Z::DOMAIN_SHIFT, bad, cls);
    return bad + cls;
}

int main() {
    int bad = check<bn254_fr_params>("bn254 fr");
    bad += check<bls12_381_fr_params>("bls12-381 fr");
    bad += check<bw6_761_fr_params>("bw6-761 fr");
    return bad > 100 ? 100 : bad;
}
