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
