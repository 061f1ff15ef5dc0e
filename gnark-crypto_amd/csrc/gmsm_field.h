// Prime-field and Fp2 arithmetic for the MSM kernels (gfx950) and for the host-side fold.
//
// Representation: N 32-bit limbs, little-endian, Montgomery form with R = 2^(32N) -- bit-identical in memory to
// the reference's `fp.Element [N/2]uint64` (ecc/bn254/fp/element.go:24-36), so Go slices are consumed as-is.
// Every result is fully reduced into [0,q), like the reference, so limb values are canonical.
//
// What each op replaces in the reference:
//   add/dbl/sub/neg   ecc/bn254/fp/element.go:386-454
//   mul/sqr           ecc/bn254/fp/element_purego.go:46 (no-carry CIOS, "Algorithm 2" of El Housni-Botrel) -- the
//                     amd64 build runs field/asm/element_4w_amd64.s:208-304 instead; here it is a 32-bit-limb CIOS
//                     whose inner step is one v_mad_u64_u32 (32x32+64 -> 64, measured 4 cycles / wave64 / SIMD)
//   from_mont         ecc/bn254/fp/element.go:593-642
//   Fp2               ecc/bn254/internal/fptower/e2_bn254.go:28-50, e2_fallback.go:10-28 (u^2 = -1)
// The no-carry variant needs the modulus' top word to leave a spare bit (field_config.go:200-206); true for all
// six fields in scope.
#pragma once
#include <stdint.h>
#include "gmsm_params32.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GMSM_HD __host__ __device__ __forceinline__
// The Montgomery product is the one big straight-line body (2N^2+N multiply-adds, fully unrolled so that limbs stay in
// VGPRs). It is a real function call on the device unless GMSM_INLINE_MUL is set: one copy in the instruction cache
// instead of ten per mixed add, and compile time that stays in seconds for the 24-limb field.
#if defined(GMSM_INLINE_MUL)
#define GMSM_MUL_HD __host__ __device__ __forceinline__
#else
#define GMSM_MUL_HD __host__ __device__ __noinline__
#endif
#else
#define GMSM_HD inline
#define GMSM_MUL_HD inline
#endif

namespace gmsm {

template <class P>
struct Fp {
    static constexpr int N = P::N;
    using Params = P;
    uint32_t l[N];

    GMSM_HD static Fp zero() {
        Fp z;
#pragma unroll
        for (int i = 0; i < N; ++i) z.l[i] = 0;
        return z;
    }
    GMSM_HD static Fp one() {
        Fp z;
#pragma unroll
        for (int i = 0; i < N; ++i) z.l[i] = P::ONE[i];
        return z;
    }
    GMSM_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc |= l[i];
        return acc == 0;
    }
    GMSM_HD bool operator==(const Fp &o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
};

// ---- host forms of the additive operations on 64-bit words. The window fold (Group::fold: (nwin - 1) c doublings in a
// serial chain after the device has finished - 0.07 ms of every MultiExp, half of a small call) spends 40 % of a doubling
// in its twelve additions / subtractions when they run as eight-word carry chains; on 64-bit words they are four steps.
// Same canonical results, limb for limb (the element's memory layout is that of [N/2]uint64, little-endian).
#if !defined(__HIP_DEVICE_COMPILE__)
template <class P>
struct FpHost64 {
    static constexpr int M = P::N / 2;
    static_assert(P::N % 2 == 0, "the host forms pack the 32-bit limbs pairwise: an odd count would drop a limb");
    static_assert((P::Q[P::N - 1] >> 31) == 0, "the host forms rely on a spare top bit of the modulus (no carry out of the top word)");
    static inline void load(const Fp<P> &x, unsigned long long *a) {
        for (int i = 0; i < M; ++i) a[i] = (unsigned long long)x.l[2 * i] | ((unsigned long long)x.l[2 * i + 1] << 32);
    }
    static inline void store(Fp<P> &z, const unsigned long long *a) {
        for (int i = 0; i < M; ++i) {
            z.l[2 * i] = (uint32_t)a[i];
            z.l[2 * i + 1] = (uint32_t)(a[i] >> 32);
        }
    }
    static inline unsigned long long q(int i) { return (unsigned long long)P::Q[2 * i] | ((unsigned long long)P::Q[2 * i + 1] << 32); }
    // add / subtract with carry on 128-bit intermediates (adc / sbb in the generated code; portable across clang and gcc -
    // the host checks under tests/c compile this header with g++)
    static inline unsigned long long addc(unsigned long long a, unsigned long long b, unsigned long long &c) {
        const unsigned __int128 t = (unsigned __int128)a + b + c;
        c = (unsigned long long)(t >> 64);
        return (unsigned long long)t;
    }
    static inline unsigned long long subb(unsigned long long a, unsigned long long b, unsigned long long &bw) {
        const unsigned __int128 t = (unsigned __int128)a - b - bw;
        bw = (unsigned long long)(t >> 64) & 1ull;
        return (unsigned long long)t;
    }
    // t < 2q -> t mod q
    static inline void reduce_once(unsigned long long *t) {
        unsigned long long d[M], b = 0;
        for (int i = 0; i < M; ++i) d[i] = subb(t[i], q(i), b);
        for (int i = 0; i < M; ++i) t[i] = b ? t[i] : d[i];
    }
};
#endif

// z = (t >= q) ? t - q : t      (t < 2q)
template <class P>
GMSM_HD void fp_reduce_once(Fp<P> &t) {
    constexpr int N = P::N;
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        using H = FpHost64<P>;
        unsigned long long a[H::M];
        H::load(t, a);
        H::reduce_once(a);
        H::store(t, a);
        return;
    }
#endif
    uint32_t d[N];
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = __builtin_subc(t.l[i], P::Q[i], b, &b);
    // b == 1  <=>  t < q  -> keep t
#pragma unroll
    for (int i = 0; i < N; ++i) t.l[i] = b ? t.l[i] : d[i];
}

template <class P>
GMSM_HD Fp<P> fp_add(const Fp<P> &x, const Fp<P> &y) {
    constexpr int N = P::N;
    Fp<P> z;
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        using H = FpHost64<P>;
        unsigned long long a[H::M], b[H::M], c = 0;
        H::load(x, a);
        H::load(y, b);
        for (int i = 0; i < H::M; ++i) a[i] = H::addc(a[i], b[i], c);
        H::reduce_once(a);  // top word of q leaves a spare bit: no carry out of the top word
        H::store(z, a);
        return z;
    }
#endif
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = __builtin_addc(x.l[i], y.l[i], c, &c);
    fp_reduce_once(z);  // top word of q leaves a spare bit: no carry out of limb N-1
    return z;
}

template <class P>
GMSM_HD Fp<P> fp_dbl(const Fp<P> &x) {
    return fp_add(x, x);
}

template <class P>
GMSM_HD Fp<P> fp_sub(const Fp<P> &x, const Fp<P> &y) {
    constexpr int N = P::N;
    Fp<P> z;
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        using H = FpHost64<P>;
        unsigned long long a[H::M], b[H::M], bw = 0, c = 0;
        H::load(x, a);
        H::load(y, b);
        for (int i = 0; i < H::M; ++i) a[i] = H::subb(a[i], b[i], bw);
        for (int i = 0; i < H::M; ++i) a[i] = H::addc(a[i], bw ? H::q(i) : 0ull, c);  // if borrow: += q
        H::store(z, a);
        return z;
    }
#endif
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = __builtin_subc(x.l[i], y.l[i], b, &b);
    // if borrow: z += q
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = __builtin_addc(z.l[i], b ? P::Q[i] : 0u, c, &c);
    return z;
}

template <class P>
GMSM_HD Fp<P> fp_neg(const Fp<P> &x) {
    constexpr int N = P::N;
    Fp<P> z;
    uint32_t b = 0;
    const bool zero = x.is_zero();
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = __builtin_subc(P::Q[i], x.l[i], b, &b);
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = zero ? 0u : z.l[i];
    return z;
}

// Montgomery product x*y*R^-1 mod q. Two interleaved carry chains per outer iteration (A for x*y_i, C for m*q),
// each inner step is a 32x32 multiply plus two 32-bit addends, which never overflows 64 bits.
template <class P>
GMSM_MUL_HD Fp<P> fp_mul(const Fp<P> x, const Fp<P> y) {
    constexpr int N = P::N;
#if !defined(__HIP_DEVICE_COMPILE__)
    // host (window fold, FromJacobian, base generator): the no-carry CIOS of the reference (element_purego.go:46-213) on
    // 64-bit limbs - two interleaved carry chains per row (A for x*y_i, C for m*q) and no extra top word, which the
    // spare top bit of every modulus in scope allows (field_config.go:200-206). 7-29 % faster than the generic CIOS
    // with a (M+1)-word accumulator it replaces (4 / 6 / 12 limbs), limb-for-limb the same results.
    {
        constexpr int M = N / 2;
        uint64_t a[M], b[M], q[M], t64[M];
        for (int i = 0; i < M; ++i) {
            a[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
            b[i] = (uint64_t)y.l[2 * i] | ((uint64_t)y.l[2 * i + 1] << 32);
            q[i] = (uint64_t)P::Q[2 * i] | ((uint64_t)P::Q[2 * i + 1] << 32);
            t64[i] = 0;
        }
        // -q^-1 mod 2^64 from the 32-bit constant by one Newton step
        uint64_t qinv = P::QINV;                     // correct mod 2^32
        qinv = qinv * (2 + q[0] * qinv);             // -q^-1 mod 2^64:  x' = x(2 + q x) for x = -q^-1
#pragma unroll
        for (int i = 0; i < M; ++i) {
            unsigned __int128 c1 = (unsigned __int128)a[0] * b[i] + t64[0];
            const uint64_t m = (uint64_t)c1 * qinv;
            unsigned __int128 c2 = (unsigned __int128)m * q[0] + (uint64_t)c1;
            uint64_t A = (uint64_t)(c1 >> 64), C = (uint64_t)(c2 >> 64);
#pragma unroll
            for (int j = 1; j < M; ++j) {
                c1 = (unsigned __int128)a[j] * b[i] + t64[j] + A;
                A = (uint64_t)(c1 >> 64);
                c2 = (unsigned __int128)m * q[j] + (uint64_t)c1 + C;
                C = (uint64_t)(c2 >> 64);
                t64[j - 1] = (uint64_t)c2;
            }
            t64[M - 1] = A + C;  // no-carry condition: cannot overflow
        }
        Fp<P> z;
        for (int i = 0; i < M; ++i) {
            z.l[2 * i] = (uint32_t)t64[i];
            z.l[2 * i + 1] = (uint32_t)(t64[i] >> 32);
        }
        fp_reduce_once(z);  // value < 2q
        return z;
    }
#endif
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t yi = y.l[i];
        uint64_t a = (uint64_t)x.l[0] * yi + t[0];
        uint32_t A = (uint32_t)(a >> 32);
        const uint32_t m = (uint32_t)a * P::QINV;
        uint64_t c = (uint64_t)m * P::Q[0] + (uint32_t)a;
        uint32_t C = (uint32_t)(c >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            a = (uint64_t)x.l[j] * yi + t[j] + A;
            A = (uint32_t)(a >> 32);
            c = (uint64_t)m * P::Q[j] + (uint32_t)a + C;
            C = (uint32_t)(c >> 32);
            t[j - 1] = (uint32_t)c;
        }
        t[N - 1] = A + C;  // no-carry condition: cannot overflow
    }
    Fp<P> z;
#pragma unroll
    for (int i = 0; i < N; ++i) z.l[i] = t[i];
    fp_reduce_once(z);
    return z;
}

template <class P>
GMSM_HD Fp<P> fp_sqr(const Fp<P> &x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host (the window fold: 5 of the 7 products of a Jacobian doubling are squares - (nwin - 1) c doublings in a serial
    // chain after the device has finished): the off-diagonal products once, doubled, plus the diagonal, then M rounds of
    // Montgomery reduction - M (M + 1) / 2 + M^2 word products instead of 2 M^2 + M (26 / 57 / 222 against 36 / 78 / 300
    // for 4 / 6 / 12 words). Same canonical result as fp_mul(x, x), word for word (tests/c/host_sqr_check.cpp).
    // Measured (tools/fold_time.py, c = 16): BN254 G1 fold 84.8 -> 61.2 us, BLS12-381 G1 159 -> 112, BN254 G2 236 -> 183; the
    // 12-word field gains nothing (871 against 875 us: 25 words of accumulator live in memory either way, and the interleaved
    // CIOS of fp_mul makes one pass over them where this makes three) and keeps fp_mul.
    if constexpr (P::N <= 12) {
        constexpr int M = P::N / 2;
        static_assert(P::N % 2 == 0, "an element is a whole number of 64-bit words");
        uint64_t a[M], q[M], t[2 * M + 1];
        for (int i = 0; i < M; ++i) {
            a[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
            q[i] = (uint64_t)P::Q[2 * i] | ((uint64_t)P::Q[2 * i + 1] << 32);
        }
        for (int i = 0; i <= 2 * M; ++i) t[i] = 0;
#pragma unroll
        for (int i = 0; i < M; ++i) {  // sum_{i < j} a_i a_j 2^(64 (i + j))
            uint64_t carry = 0;
#pragma unroll
            for (int j = i + 1; j < M; ++j) {
                const unsigned __int128 c = (unsigned __int128)a[i] * a[j] + t[i + j] + carry;
                t[i + j] = (uint64_t)c;
                carry = (uint64_t)(c >> 64);
            }
            t[i + M] = carry;
        }
        {  // 2 * that + sum_i a_i^2 2^(128 i)
            uint64_t top = 0, carry = 0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const unsigned __int128 d = (unsigned __int128)a[i] * a[i];
                const uint64_t lo2 = (t[2 * i] << 1) | top, hi2 = (t[2 * i + 1] << 1) | (t[2 * i] >> 63);
                top = t[2 * i + 1] >> 63;
                unsigned __int128 s0 = (unsigned __int128)lo2 + (uint64_t)d + carry;
                t[2 * i] = (uint64_t)s0;
                unsigned __int128 s1 = (unsigned __int128)hi2 + (uint64_t)(d >> 64) + (uint64_t)(s0 >> 64);
                t[2 * i + 1] = (uint64_t)s1;
                carry = (uint64_t)(s1 >> 64);
            }
        }
        uint64_t qinv = P::QINV;             // -q^-1 mod 2^32 ...
        qinv = qinv * (2 + q[0] * qinv);     // ... mod 2^64 by one Newton step (as in fp_mul)
#pragma unroll
        for (int i = 0; i < M; ++i) {  // t += m q 2^(64 i): word i becomes zero
            const uint64_t m = t[i] * qinv;
            uint64_t carry = 0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const unsigned __int128 c = (unsigned __int128)m * q[j] + t[i + j] + carry;
                t[i + j] = (uint64_t)c;
                carry = (uint64_t)(c >> 64);
            }
            for (int k = i + M; carry != 0 && k <= 2 * M; ++k) {
                const unsigned __int128 c = (unsigned __int128)t[k] + carry;
                t[k] = (uint64_t)c;
                carry = (uint64_t)(c >> 64);
            }
        }
        Fp<P> z;  // (x^2 + (sum m_i 2^(64 i)) q) / 2^(64 M) < 2q < 2^(64 M): t[2 M] is zero
        for (int i = 0; i < M; ++i) {
            z.l[2 * i] = (uint32_t)t[M + i];
            z.l[2 * i + 1] = (uint32_t)(t[M + i] >> 32);
        }
        fp_reduce_once(z);
        return z;
    }
#endif
    return fp_mul(x, x);
}

// x * R^-1 mod q: N rounds of (z + m q) / 2^32
template <class P>
GMSM_HD Fp<P> fp_from_mont(const Fp<P> &x) {
    constexpr int N = P::N;
    Fp<P> z = x;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const uint32_t m = z.l[0] * P::QINV;
        uint64_t c = (uint64_t)m * P::Q[0] + z.l[0];
        uint32_t C = (uint32_t)(c >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c = (uint64_t)m * P::Q[j] + z.l[j] + C;
            C = (uint32_t)(c >> 32);
            z.l[j - 1] = (uint32_t)c;
        }
        z.l[N - 1] = C;
    }
    fp_reduce_once(z);
    return z;
}

// x^-1 = x^(q-2) (value-identical to the reference's bingcd Inverse; 0 -> 0). Host-side use only (final
// FromJacobian, ecc/bn254/g1.go:150-166); not on the device hot path.
template <class P>
GMSM_HD Fp<P> fp_inv(const Fp<P> &x) {
    constexpr int N = P::N;
    if (x.is_zero()) return x;
    uint32_t e[N];
    uint32_t b = 2;
    for (int i = 0; i < N; ++i) {
        uint32_t old = P::Q[i];
        e[i] = old - b;
        b = old < b ? 1u : 0u;
    }
    int top = N * 32 - 1;
    while (!((e[top / 32] >> (top % 32)) & 1)) --top;
    Fp<P> acc = x;
    for (int i = top - 1; i >= 0; --i) {
        acc = fp_sqr(acc);
        if ((e[i / 32] >> (i % 32)) & 1) acc = fp_mul(acc, x);
    }
    return acc;
}

// ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2+1)
template <class P>
struct Fp2 {
    using Params = P;
    static constexpr int N = 2 * P::N;  // 32-bit words per element
    Fp<P> a0, a1;
    GMSM_HD static Fp2 zero() { return Fp2{Fp<P>::zero(), Fp<P>::zero()}; }
    GMSM_HD static Fp2 one() { return Fp2{Fp<P>::one(), Fp<P>::zero()}; }
    GMSM_HD bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    GMSM_HD bool operator==(const Fp2 &o) const { return a0 == o.a0 && a1 == o.a1; }
};

template <class P> GMSM_HD Fp2<P> fp_add(const Fp2<P> &x, const Fp2<P> &y) { return Fp2<P>{fp_add(x.a0, y.a0), fp_add(x.a1, y.a1)}; }
template <class P> GMSM_HD Fp2<P> fp_sub(const Fp2<P> &x, const Fp2<P> &y) { return Fp2<P>{fp_sub(x.a0, y.a0), fp_sub(x.a1, y.a1)}; }
template <class P> GMSM_HD Fp2<P> fp_dbl(const Fp2<P> &x) { return Fp2<P>{fp_dbl(x.a0), fp_dbl(x.a1)}; }
template <class P> GMSM_HD Fp2<P> fp_neg(const Fp2<P> &x) { return Fp2<P>{fp_neg(x.a0), fp_neg(x.a1)}; }

template <class P>
GMSM_HD Fp2<P> fp_mul(const Fp2<P> &x, const Fp2<P> &y) {  // Karatsuba, 3 base muls
    Fp<P> a = fp_add(x.a0, x.a1);
    Fp<P> b = fp_add(y.a0, y.a1);
    a = fp_mul(a, b);
    b = fp_mul(x.a0, y.a0);
    Fp<P> c = fp_mul(x.a1, y.a1);
    Fp2<P> z;
    z.a1 = fp_sub(fp_sub(a, b), c);
    z.a0 = fp_sub(b, c);
    return z;
}

template <class P>
GMSM_HD Fp2<P> fp_sqr(const Fp2<P> &x) {  // 2 base muls
    Fp<P> a = fp_add(x.a0, x.a1);
    Fp<P> b = fp_sub(x.a0, x.a1);
    a = fp_mul(a, b);
    b = fp_dbl(fp_mul(x.a0, x.a1));
    return Fp2<P>{a, b};
}

template <class P>
GMSM_HD Fp2<P> fp_inv(const Fp2<P> &x) {
    Fp<P> t0 = fp_add(fp_sqr(x.a0), fp_sqr(x.a1));
    Fp<P> t1 = fp_inv(t0);
    return Fp2<P>{fp_mul(x.a0, t1), fp_neg(fp_mul(x.a1, t1))};
}

}  // namespace gmsm
