// Host-side field code of the window fold: the dedicated squaring of gmsm_field.h (off-diagonal products once, doubled) against
// fp_mul(x, x), on random values and on the edges (0, 1, q - 1, all words set below q) for the six fields in scope.
//   g++ -O2 -std=c++17 -D__host__= -D__device__= -D__noinline__= -o host_sqr_check tests/c/host_sqr_check.cpp && ./host_sqr_check
#include <cstdio>
#include <cstdint>
#include <random>
#include "../../gnark-crypto_amd/csrc/gmsm_field.h"
using namespace gmsm;

template <class P>
static int check(const char *name) {
    std::mt19937_64 rng(12345);
    int bad = 0;
    auto one = [&](Fp<P> x) {
        const Fp<P> a = fp_sqr(x), b = fp_mul(x, x);
        if (!(a == b)) ++bad;
    };
    Fp<P> e = Fp<P>::zero();
    one(e);
    one(Fp<P>::one());
    for (int i = 0; i < P::N; ++i) e.l[i] = P::Q[i];
    e.l[0] -= 1;  // q - 1 (q is odd)
    one(e);
    for (int i = 0; i < P::N; ++i) e.l[i] = 0xffffffffu;
    e.l[P::N - 1] = P::Q[P::N - 1] - 1;  // below q, every other word saturated
    one(e);
    for (int it = 0; it < 200000; ++it) {
        Fp<P> x;
        for (int i = 0; i < P::N; ++i) x.l[i] = (uint32_t)rng();
        x.l[P::N - 1] %= P::Q[P::N - 1];  // < q
        one(x);
        // chains: what the fold does
        Fp<P> y = x;
        for (int k = 0; k < 3; ++k) {
            const Fp<P> s1 = fp_sqr(y), s2 = fp_mul(y, y);
            if (!(s1 == s2)) ++bad;
            y = s1;
        }
    }
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}

int main() {
    int bad = 0;
    bad += check<bn254_fp_params>("bn254 fp");
    bad += check<bn254_fr_params>("bn254 fr");
    bad += check<bls12_381_fp_params>("bls12_381 fp");
    bad += check<bls12_381_fr_params>("bls12_381 fr");
    bad += check<bw6_761_fp_params>("bw6_761 fp");
    bad += check<bw6_761_fr_params>("bw6_761 fr");
    return bad ? 1 : 0;
}
