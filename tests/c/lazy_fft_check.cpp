// Host-side check of the lazy-limb FFT butterflies (gmsm_fft_lazy.h) against the saturated canonical field
// (gmsm_field.h = what the reference computes). The device runs every transform with Cooley-Tukey butterflies and no
// reduction inside a pass of up to DIT_FREE_STAGES stages (FftLz::dit_one / dit_free), in two orders:
//   bottom-up (decimation DIT): bit 0 first, twiddle w^(j << (log n - 1 - b)), j = i mod 2^b  - the reference's ditFFT;
//   top-down  (decimation DIF): bit log n - 1 first, twiddle w^(bitrev_(log n - 1)(i >> (b + 1))) - computes what the
//   reference's Gentleman-Sande difFFT computes (fft.go:198-262), checked here against exactly that recursion.
// Full 2^11-point transforms in one chain (the longest a device pass runs), started from the WORST input the device ever
// loads - a lazy representative just below 2^(32N): the canonical value plus as many q as fit - on random inputs and on
// inputs that maximise growth (all q - 1, all zero, alternating). Every intermediate must stay below 40q with nearly
// normalised limbs; the results must come out canonical and equal (store_big), also through the lazy store + reload.
// Exit code = number of mismatches (capped).
// Build: clang++ -O2 -std=c++17 -D__host__= -D__device__= -D__noinline__= tests/c/lazy_fft_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../gnark-crypto_amd/csrc/gmsm_params32.h"
#include "../../gnark-crypto_amd/csrc/gmsm_fft_lazy.h"
using namespace gmsm;

static size_t bitrev(size_t v, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

template <class P>
static int check(const char *name) {
    using Fr = Fp<P>;
    using Z = FftLz<P>;
    constexpr unsigned LOG = Z::DIT_FREE_STAGES;
    constexpr size_t N = (size_t)1 << LOG;
    std::mt19937_64 g(0xf17);
    Fr root;
    for (int i = 0; i < P::N; ++i) root.l[i] = P::ROOT_OF_UNITY[i];
    for (unsigned i = 0; i < P::MAX_ORDER - LOG; ++i) root = fp_sqr(root);
    Fr shift = Fr::one();
    for (unsigned i = 0; i < Z::DOMAIN_SHIFT; ++i) shift = fp_dbl(shift);
    std::vector<Fr> tw(N / 2), twz(N / 2), twr(N / 2);
    tw[0] = Fr::one();
    for (size_t t = 1; t < N / 2; ++t) tw[t] = fp_mul(tw[t - 1], root);
    for (size_t t = 0; t < N / 2; ++t) twz[t] = fp_mul(tw[t], shift);
    for (size_t k = 0; k < N / 2; ++k) twr[k] = twz[bitrev(k, LOG - 1)];  // the bit-reversed table of the top-down passes
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
        for (int topdown = 0; topdown < 2; ++topdown) {
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
                const unsigned b = topdown ? LOG - 1 - st : st;  // bit paired by this stage
                for (size_t q = 0; q < N / 2; ++q) {
                    const size_t i0 = ((q >> b) << (b + 1)) | (q & (((size_t)1 << b) - 1)), i1 = i0 | ((size_t)1 << b);
                    const size_t j = i0 & (((size_t)1 << b) - 1), t = j << (LOG - 1 - b);
                    if (topdown) {  // the reference's Gentleman-Sande stage on the saturated field
                        const Fr d = fp_mul(fp_sub(s[i0], s[i1]), tw[t]);
                        s[i0] = fp_add(s[i0], s[i1]);
                        s[i1] = d;
                    } else {        // the reference's Cooley-Tukey stage
                        const Fr tt = fp_mul(s[i1], tw[t]);
                        s[i1] = fp_sub(s[i0], tt);
                        s[i0] = fp_add(s[i0], tt);
                    }
                    const bool carry = ((LOG - 1 - st) & 1u) != 0;  // as the kernel: every second stage, never the last
                    for (const FpU<P> *v : {&z[i0], &z[i1]})    // what a stage may be handed: limbs <= 4 * 2^W + 8
                        for (int k = 0; k < P::UL - 1; ++k)
                            if (v->l[k] > (4u << P::UW) + 8u) ++cls;
                    if (b == (topdown ? LOG - 1 : 0u)) Z::dit_one(z[i0], z[i1]);
                    else if (carry) Z::template dit_free<true>(z[i0], z[i1], topdown ? twr[i0 >> (b + 1)] : twz[t]);
                    else Z::template dit_free<false>(z[i0], z[i1], topdown ? twr[i0 >> (b + 1)] : twz[t]);
                    for (const FpU<P> *v : {&z[i0], &z[i1]}) {
                        if (carry || b == (topdown ? LOG - 1 : 0u))
                            for (int k = 0; k < P::UL - 1; ++k)
                                if (v->l[k] > (1u << P::UW) + (1u << (32 - P::UW))) ++cls;
                        FpU<P> nv = *v;
                        fpu_normalize(nv);
                        if ((uint64_t)nv.l[P::UL - 1] >= 40ull * (P::UQ1[P::UL - 1] + 1ull)) ++cls;  // below 40q
                    }
                }
            }
            // bottom-up: the lazy chain mirrors the reference stage by stage, so the results agree element by element.
            // top-down: the two flow graphs differ stage by stage and agree at the END (both hold the DFT in bit-reversed order).
            for (size_t i = 0; i < N; ++i) {
                const Fr o = Z::store_big(z[i]);
                if (memcmp(&o, &s[i], sizeof o) != 0 && bad++ < 5) printf("%s: pattern %d topdown %d: element %zu differs\n", name, pattern, topdown, i);
                const Fr lz = Z::store_lazy_big(z[i]);  // what a pass that is not the last stores; the next pass re-cuts it
                const Fr back = Z::store_big(Z::load(lz));
                if (memcmp(&back, &s[i], sizeof back) != 0 && bad++ < 5) printf("%s: pattern %d topdown %d lazy store: element %zu differs\n", name, pattern, topdown, i);
                const Fr viaprod = Z::store(Z::mul(z[i], shift));  // the last pass of an inverse transform ends in a product
                const Fr ref = fp_mul(s[i], Fr::one());
                (void)ref;
                const Fr expect = s[i];
                if (memcmp(&viaprod, &expect, sizeof expect) != 0 && bad++ < 5) printf("%s: pattern %d topdown %d product store: element %zu differs\n", name, pattern, topdown, i);
            }
        }
    }
    printf("%s: shift=%u: %d mismatches, %d class violations\n", name, Z::DOMAIN_SHIFT, bad, cls);
    return bad + cls;
}

int main() {
    int bad = check<bn254_fr_params>("bn254 fr");
    bad += check<bls12_381_fr_params>("bls12-381 fr");
    bad += check<bw6_761_fr_params>("bw6-761 fr");
    return bad > 100 ? 100 : bad;
}
