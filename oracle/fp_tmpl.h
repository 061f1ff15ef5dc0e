/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product path.
 *
 * Prime-field template: a CPU restatement of gnark-crypto's generated Montgomery field arithmetic.
 * Instantiate by defining, then including this file:
 *   FP          identifier prefix (e.g. bn254_fp)
 *   FP_N        number of 64-bit limbs
 *   FP_Q        modulus limb array            (const uint64_t[FP_N])
 *   FP_QINVNEG  -q^-1 mod 2^64
 *   FP_ONE      R mod q limb array
 *   FP_RSQ      R^2 mod q limb array
 *
 * Follows (4-limb instance cited; the 6- and 12-limb generated twins are line-for-line analogous):
 *   Add/Double/Sub/Neg        ecc/bn254/fp/element.go:386-454
 *   Mul (textbook CIOS)       ecc/bn254/fp/element.go:470-591 (_mulGeneric) with madd0/1/2 of fp/arith.go:12-47
 *   fromMont                  ecc/bn254/fp/element.go:593-642 (_fromMontGeneric)
 *   smallerThanModulus        ecc/bn254/fp/element.go:345-352
 *   Inverse (x^(q-2) variant) ecc/bn254/fp/element.go:1327-1343 (inverseExp; same canonical value as the bingcd)
 * All results are fully reduced into [0,q), like the reference.
 */
#include <stdint.h>
#include <string.h>

#ifndef ORACLE_CAT
#define ORACLE_CAT_(a, b) a##b
#define ORACLE_CAT(a, b) ORACLE_CAT_(a, b)
typedef unsigned __int128 oracle_u128;
#endif

#define FPT ORACLE_CAT(FP, _t)
#define FPF(name) ORACLE_CAT(FP, ORACLE_CAT(_, name))

typedef struct { uint64_t l[FP_N]; } FPT;

static inline int FPF(is_zero)(const FPT *x) {
    uint64_t acc = 0;
    for (int i = 0; i < FP_N; ++i) acc |= x->l[i];
    return acc == 0;
}

static inline int FPF(equal)(const FPT *x, const FPT *y) {
    uint64_t acc = 0;
    for (int i = 0; i < FP_N; ++i) acc |= x->l[i] ^ y->l[i];
    return acc == 0;
}

static inline void FPF(set_zero)(FPT *z) { memset(z, 0, sizeof *z); }
static inline void FPF(set_one)(FPT *z) { memcpy(z->l, FP_ONE, sizeof z->l); }

/* x < q ?  (element.go:345 smallerThanModulus: lexicographic from the top limb) */
static inline int FPF(lt_modulus)(const uint64_t *x) {
    for (int i = FP_N - 1; i >= 0; --i) {
        if (x[i] < FP_Q[i]) return 1;
        if (x[i] > FP_Q[i]) return 0;
    }
    return 0; /* equal */
}

static inline void FPF(sub_q)(uint64_t *z) {
    uint64_t b = 0;
    for (int i = 0; i < FP_N; ++i) {
        oracle_u128 d = (oracle_u128)z[i] - FP_Q[i] - b;
        z[i] = (uint64_t)d;
        b = (uint64_t)(d >> 64) & 1;
    }
}

/* z = x + y mod q  (element.go:386) -- the moduli in scope leave a spare top bit, so the carry out of the
 * top limb is always 0, as the generated code assumes. */
static inline void FPF(add)(FPT *z, const FPT *x, const FPT *y) {
    uint64_t c = 0;
    for (int i = 0; i < FP_N; ++i) {
        oracle_u128 s = (oracle_u128)x->l[i] + y->l[i] + c;
        z->l[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
    }
    if (!FPF(lt_modulus)(z->l)) FPF(sub_q)(z->l);
}

static inline void FPF(dbl)(FPT *z, const FPT *x) { FPF(add)(z, x, x); } /* element.go:406 */

/* z = x - y mod q  (element.go:426) */
static inline void FPF(sub)(FPT *z, const FPT *x, const FPT *y) {
    uint64_t b = 0;
    for (int i = 0; i < FP_N; ++i) {
        oracle_u128 d = (oracle_u128)x->l[i] - y->l[i] - b;
        z->l[i] = (uint64_t)d;
        b = (uint64_t)(d >> 64) & 1;
    }
    if (b) {
        uint64_t c = 0;
        for (int i = 0; i < FP_N; ++i) {
            oracle_u128 s = (oracle_u128)z->l[i] + FP_Q[i] + c;
            z->l[i] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
    }
}

/* z = q - x, 0 -> 0  (element.go:443) */
static inline void FPF(neg)(FPT *z, const FPT *x) {
    if (FPF(is_zero)(x)) { FPF(set_zero)(z); return; }
    uint64_t b = 0;
    for (int i = 0; i < FP_N; ++i) {
        oracle_u128 d = (oracle_u128)FP_Q[i] - x->l[i] - b;
        z->l[i] = (uint64_t)d;
        b = (uint64_t)(d >> 64) & 1;
    }
}

/* z = x*y*R^-1 mod q: CIOS, one outer iteration per limb of y (element.go:470-591).
 * t has N+1 limbs plus the overflow bit D, exactly as _mulGeneric.  Kept as the cross-check of FPF(mul) below
 * (tests/test_oracle_pinning.py compares the two on random and edge values). */
static inline void FPF(mul_generic)(FPT *z, const FPT *x, const FPT *y) {
    uint64_t t[FP_N + 1];
    memset(t, 0, sizeof t);
    for (int i = 0; i < FP_N; ++i) {
        /* first loop: t += x * y[i] */
        uint64_t C = 0;
        for (int j = 0; j < FP_N; ++j) {
            oracle_u128 p = (oracle_u128)y->l[i] * x->l[j] + t[j] + C; /* madd2 */
            t[j] = (uint64_t)p;
            C = (uint64_t)(p >> 64);
        }
        oracle_u128 s = (oracle_u128)t[FP_N] + C;
        t[FP_N] = (uint64_t)s;
        uint64_t D = (uint64_t)(s >> 64);
        /* m = t[0] * q' mod W */
        uint64_t m = t[0] * FP_QINVNEG;
        /* second loop: t = (t + m*q) / W */
        oracle_u128 p0 = (oracle_u128)m * FP_Q[0] + t[0]; /* madd0: low word is 0 by construction */
        C = (uint64_t)(p0 >> 64);
        for (int j = 1; j < FP_N; ++j) {
            oracle_u128 p = (oracle_u128)m * FP_Q[j] + t[j] + C; /* madd2 */
            t[j - 1] = (uint64_t)p;
            C = (uint64_t)(p >> 64);
        }
        s = (oracle_u128)t[FP_N] + C;
        t[FP_N - 1] = (uint64_t)s;
        t[FP_N] = D + (uint64_t)(s >> 64);
    }
    /* moduli in scope: t[N] == 0 here (spare bits), result < 2q */
    for (int i = 0; i < FP_N; ++i) z->l[i] = t[i];
    if (t[FP_N] || !FPF(lt_modulus)(z->l)) FPF(sub_q)(z->l);
}

/* z = x*y*R^-1 mod q, the "no-carry" interleaving the reference's Mul uses for these moduli (top word of q below
 * 2^63 and not all ones): element_purego.go:46-213 and, as MULX/ADCX/ADOX assembly, field/asm/element_4w_amd64.s:208-304.
 * One pass per limb of y merges the product row and the reduction row, N words of state, no overflow word.  Fully
 * unrolled; with -mbmi2 -madx gcc emits MULX and ADC chains for the 128-bit expressions. */
static inline void FPF(mul)(FPT *z, const FPT *x, const FPT *y) {
    uint64_t t[FP_N];
#pragma GCC unroll 16
    for (int j = 0; j < FP_N; ++j) t[j] = 0;
#pragma GCC unroll 16
    for (int i = 0; i < FP_N; ++i) {
        const uint64_t yi = y->l[i];
        oracle_u128 a = (oracle_u128)x->l[0] * yi + t[0];           /* (A, t0) = t0 + x0*yi */
        const uint64_t m = (uint64_t)a * FP_QINVNEG;
        oracle_u128 c = (oracle_u128)m * FP_Q[0] + (uint64_t)a;      /* (C, _) = t0 + m*q0 */
        uint64_t A = (uint64_t)(a >> 64), C = (uint64_t)(c >> 64);
#pragma GCC unroll 16
        for (int j = 1; j < FP_N; ++j) {
            a = (oracle_u128)x->l[j] * yi + t[j] + A;                /* (A, tj) = tj + xj*yi + A */
            A = (uint64_t)(a >> 64);
            c = (oracle_u128)m * FP_Q[j] + (uint64_t)a + C;          /* (C, t[j-1]) = tj + m*qj + C */
            t[j - 1] = (uint64_t)c;
            C = (uint64_t)(c >> 64);
        }
        t[FP_N - 1] = C + A;
    }
#pragma GCC unroll 16
    for (int i = 0; i < FP_N; ++i) z->l[i] = t[i];
    if (!FPF(lt_modulus)(z->l)) FPF(sub_q)(z->l);
}

static inline void FPF(sqr)(FPT *z, const FPT *x) { FPF(mul)(z, x, x); } /* Square == Mul(x,x) value-wise */

/* z = z * R^-1 mod q  (element.go:593 _fromMontGeneric: N rounds of "z = (z + m q)/W") */
static inline void FPF(from_mont)(FPT *z) {
    for (int r = 0; r < FP_N; ++r) {
        uint64_t m = z->l[0] * FP_QINVNEG;
        oracle_u128 p0 = (oracle_u128)m * FP_Q[0] + z->l[0];
        uint64_t C = (uint64_t)(p0 >> 64);
        for (int j = 1; j < FP_N; ++j) {
            oracle_u128 p = (oracle_u128)m * FP_Q[j] + z->l[j] + C;
            z->l[j - 1] = (uint64_t)p;
            C = (uint64_t)(p >> 64);
        }
        z->l[FP_N - 1] = C;
    }
    if (!FPF(lt_modulus)(z->l)) FPF(sub_q)(z->l);
}

/* z = x*R mod q (Montgomery form of the plain integer x < q) */
static inline void FPF(to_mont)(FPT *z, const FPT *x) {
    FPT rsq;
    memcpy(rsq.l, FP_RSQ, sizeof rsq.l);
    FPF(mul)(z, x, &rsq);
}

/* z = x^-1 via x^(q-2) (element.go:1327 inverseExp). 0 -> 0 like the reference (Inverse of 0 is 0). */
static inline void FPF(inv)(FPT *z, const FPT *x) {
    if (FPF(is_zero)(x)) { FPF(set_zero)(z); return; }
    uint64_t e[FP_N];
    memcpy(e, FP_Q, sizeof e);
    /* e = q - 2 (q is odd and > 2, so only the low limb can borrow) */
    uint64_t b = 2;
    for (int i = 0; i < FP_N && b; ++i) {
        uint64_t old = e[i];
        e[i] = old - b;
        b = old < b ? 1 : 0;
    }
    int top = FP_N * 64 - 1;
    while (!((e[top / 64] >> (top % 64)) & 1)) --top;
    FPT acc = *x, base = *x;
    for (int i = top - 1; i >= 0; --i) {
        FPF(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) FPF(mul)(&acc, &acc, &base);
    }
    *z = acc;
}

#undef FPT
#undef FPF
