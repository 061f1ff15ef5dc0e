// Host-side check of the lazy-limb field code the kernels are built from (gmsm_fieldu.h / gmsm_field2u.h are plain
// C++ under the HIP qualifiers): the double product with one Montgomery reduction (fpu_mul_add) against two reduced
// products, and the Fp2 multiplication built on it against the Karatsuba form, on random members of the reduced class
// [0, 4q) and on the edges of that class (0, 4q - 1, q, 2q, 3q). Exit code = number of mismatches (capped).
// Build: g++ -O2 -std=c++17 -D__host__= -D__device__= -D__noinline__= tests/c/lazy_field_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../gnark-crypto_amd/csrc/gmsm_params32.h"
#include "../../gnark-crypto_amd/csrc/gmsm_field2u.h"
using namespace gmsm;

template <class P>
static FpU<P> member_of_R(std::mt19937_64 &g, int edge) {
    FpU<P> r;
    if (edge >= 0) {  // k*q, or 4q - 1
        for (int i = 0; i < P::UL; ++i) r.l[i] = edge == 4 ? P::UQ4[i] : edge == 3 ? P::UQ3[i] : edge == 2 ? P::UQ2[i] : edge == 1 ? P::UQ1[i] : 0u;
        if (edge == 4) r.l[0] -= 1;
        return r;
    }
    Fp<P> x;
    for (int i = 0; i < P::N; ++i) x.l[i] = (uint32_t)g();
    x.l[P::N - 1] %= P::Q[P::N - 1];  // below q
    r = fpu_unpack<P>(x.l);
    const int k = (int)(g() & 3);  // + k q, still below 4q
    for (int t = 0; t < k; ++t) {
        for (int i = 0; i < P::UL; ++i) r.l[i] += P::UQ[i];
        fpu_normalize(r);
    }
    return r;
}

template <class P>
static bool same_residue(const FpU<P> &a, const FpU<P> &b) {
    const Fp<P> x = fpu_to_sat<P, true>(a), y = fpu_to_sat<P, true>(b);
    return memcmp(&x, &y, sizeof x) == 0;
}

template <class P>
static int check(const char *name, int iters) {
    std::mt19937_64 g(20260925);
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        const bool edges = it < 625;  // all 5^4 edge combinations first
        auto pick = [&](int slot) { return member_of_R<P>(g, edges ? (it / (slot == 0 ? 1 : slot == 1 ? 5 : slot == 2 ? 25 : 125)) % 5 : -1); };
        const Fp2U<P> x{pick(0), pick(1)}, y{pick(2), pick(3)};
        const Fp2U<P> z = lz_mul<true>(x, y);
        const FpU<P> t0 = fpu_mul(x.a0, y.a0), t1 = fpu_mul(x.a1, y.a1);
        const FpU<P> t2 = fpu_mul(fpu_add_raw(x.a0, x.a1), fpu_add_raw(y.a0, y.a1));
        const FpU<P> k0 = fpu_subr(t0, t1), k1 = fpu_subr(fpu_subr(t2, t0), t1);
        bool ok = same_residue<P>(z.a0, k0) && same_residue<P>(z.a1, k1);
        for (int i = 0; i < P::UL - 1; ++i) ok = ok && !(z.a0.l[i] >> P::UW) && !(z.a1.l[i] >> P::UW);  // normalised limbs
        // result below 2q (in R): z - 2q must be negative
        FpU<P> twoq;
        for (int i = 0; i < P::UL; ++i) twoq.l[i] = P::UQ2[i];
        ok = ok && (z.a0.l[P::UL - 1] <= twoq.l[P::UL - 1]) && (z.a1.l[P::UL - 1] <= twoq.l[P::UL - 1]);
        // prime-field double product with the operand classes of madd_u: (a b + c d) vs mul + mul + add
        const FpU<P> m1 = fpu_mul_add(x.a0, y.a0, x.a1, y.a1);
        const FpU<P> m2 = fpu_add(fpu_mul(x.a0, y.a0), fpu_mul(x.a1, y.a1));
        ok = ok && same_residue<P>(m1, m2);
        if (!ok && bad++ < 5) printf("%s: mismatch at iteration %d\n", name, it);
    }
    printf("%s: %d iterations, %d mismatches\n", name, iters, bad);
    return bad;
}

int main() {
    int bad = check<bn254_fp_params>("bn254 fp", 100000);
    bad += check<bls12_381_fp_params>("bls12-381 fp", 60000);
    bad += check<bw6_761_fp_params>("bw6-761 fp", 15000);
    return bad > 100 ? 100 : bad;
}
