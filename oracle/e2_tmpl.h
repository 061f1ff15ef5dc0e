/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product path.
 *
 * Quadratic extension Fp2 = Fp[u]/(u^2+1) template (G2 coordinates of BN254 and BLS12-381).
 * Define before including:  E2 (prefix, e.g. bn254_e2)  and  BF (base-field prefix, e.g. bn254_fp).
 *
 * Follows: type E2{A0,A1}          ecc/bn254/internal/fptower/e2.go:14-16
 *          add/sub/double/neg      ecc/bn254/internal/fptower/e2_fallback.go:10-28
 *          mulGenericE2 (Karatsuba, u^2=-1)   e2_bn254.go:28-37   (BLS12-381: e2_bls381.go:15-24)
 *          squareGenericE2         e2_bn254.go:41-50
 *          Inverse                 e2_bn254.go:61-72
 * Memory layout = Go's: A0 limbs then A1 limbs.
 */
#define E2T ORACLE_CAT(E2, _t)
#define E2F(name) ORACLE_CAT(E2, ORACLE_CAT(_, name))
#define BFT ORACLE_CAT(BF, _t)
#define BFF(name) ORACLE_CAT(BF, ORACLE_CAT(_, name))

typedef struct { BFT a0, a1; } E2T;

static inline int E2F(is_zero)(const E2T *x) { return BFF(is_zero)(&x->a0) && BFF(is_zero)(&x->a1); }
static inline int E2F(equal)(const E2T *x, const E2T *y) { return BFF(equal)(&x->a0, &y->a0) && BFF(equal)(&x->a1, &y->a1); }
static inline void E2F(set_zero)(E2T *z) { BFF(set_zero)(&z->a0); BFF(set_zero)(&z->a1); }
static inline void E2F(set_one)(E2T *z) { BFF(set_one)(&z->a0); BFF(set_zero)(&z->a1); }
static inline void E2F(add)(E2T *z, const E2T *x, const E2T *y) { BFF(add)(&z->a0, &x->a0, &y->a0); BFF(add)(&z->a1, &x->a1, &y->a1); }
static inline void E2F(sub)(E2T *z, const E2T *x, const E2T *y) { BFF(sub)(&z->a0, &x->a0, &y->a0); BFF(sub)(&z->a1, &x->a1, &y->a1); }
static inline void E2F(dbl)(E2T *z, const E2T *x) { BFF(dbl)(&z->a0, &x->a0); BFF(dbl)(&z->a1, &x->a1); }
static inline void E2F(neg)(E2T *z, const E2T *x) { BFF(neg)(&z->a0, &x->a0); BFF(neg)(&z->a1, &x->a1); }

static inline void E2F(mul)(E2T *z, const E2T *x, const E2T *y) {
    BFT a, b, c;
    BFF(add)(&a, &x->a0, &x->a1);
    BFF(add)(&b, &y->a0, &y->a1);
    BFF(mul)(&a, &a, &b);
    BFF(mul)(&b, &x->a0, &y->a0);
    BFF(mul)(&c, &x->a1, &y->a1);
    BFF(sub)(&z->a1, &a, &b);
    BFF(sub)(&z->a1, &z->a1, &c);
    BFF(sub)(&z->a0, &b, &c);
}

static inline void E2F(sqr)(E2T *z, const E2T *x) {
    BFT a, b;
    BFF(add)(&a, &x->a0, &x->a1);
    BFF(sub)(&b, &x->a0, &x->a1);
    BFF(mul)(&a, &a, &b);
    BFF(mul)(&b, &x->a0, &x->a1);
    BFF(dbl)(&b, &b);
    z->a0 = a;
    z->a1 = b;
}

static inline void E2F(inv)(E2T *z, const E2T *x) {
    BFT t0, t1;
    BFF(sqr)(&t0, &x->a0);
    BFF(sqr)(&t1, &x->a1);
    BFF(add)(&t0, &t0, &t1);
    BFF(inv)(&t1, &t0);
    BFF(mul)(&z->a0, &x->a0, &t1);
    BFF(mul)(&z->a1, &x->a1, &t1);
    BFF(neg)(&z->a1, &z->a1);
}

#undef E2T
#undef E2F
#undef BFT
#undef BFF
