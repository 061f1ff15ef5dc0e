/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product path.
 *
 * Curve group + MultiExp template: CPU restatement of gnark-crypto's Pippenger MSM for one (curve, group).
 * Define before including:
 *   G        prefix, e.g. bn254_g1
 *   CF       coordinate-field prefix (an fp_tmpl.h or e2_tmpl.h instance)
 *   SF       scalar-field prefix (an fp_tmpl.h instance), SF_N its limb count, SF_BITS its bit length
 *   G_CS     initializer list of the window sizes the reference's bestC may pick (multiexp.go:77)
 *
 * Follows (BN254 G1 cited; G2 and the other curves are generated from the same templates):
 *   g1JacExtended ops  ecc/bn254/g1.go:682-985    (SetInfinity :688, add :736, double :795, addMixed :822,
 *                                                  subMixed :878, doubleNegMixed :933, doubleMixed :962)
 *   unsafeFromJacExtended :726, G1Jac.AddAssign :243, DoubleAssign :396, FromJacobian :150
 *   MultiExp driver    ecc/bn254/multiexp.go:32-146 (bestC :75-93, split :98-140)
 *   _innerMsmG1        multiexp.go:148-209          (one worker per window; the >=115% split is a scheduling
 *                                                    detail that does not change the group element)
 *   partitionScalars   multiexp.go:709-803, computeNbChunks :681, lastC :690
 *   processChunkG1Jacobian  multiexp_jacobian.go:8-61
 *   msmReduceChunkG1Affine  multiexp.go:302-315
 *   processChunkG1BatchAffine  multiexp_affine.go:24-231, batchAddG1Affine g1.go:1122-1182, selected per window by
 *                              getChunkProcessorG1's thresholds (multiexp.go:213-299: c >= 10 and at least batchSize
 *                              buckets hit, batchSize 80/150/200/350/400/500/640 for c = 10..16)
 * oracle_set_batch_affine(0) forces the extended-Jacobian buckets everywhere (the two must give the same point:
 * multiexp_test.go:221-272).
 */
#include <stdlib.h>
#include <pthread.h>
#include <math.h>

#define GF(name) ORACLE_CAT(G, ORACLE_CAT(_, name))
#define CFT ORACLE_CAT(CF, _t)
#define CFF(name) ORACLE_CAT(CF, ORACLE_CAT(_, name))
#define SFT ORACLE_CAT(SF, _t)
#define SFF(name) ORACLE_CAT(SF, ORACLE_CAT(_, name))
#define AFF ORACLE_CAT(G, _aff_t)
#define JAC ORACLE_CAT(G, _jac_t)
#define XYZZ ORACLE_CAT(G, _xyzz_t)

typedef struct { CFT x, y; } AFF;
typedef struct { CFT x, y, z; } JAC;
typedef struct { CFT x, y, zz, zzz; } XYZZ;

/* ---------------------------------------------------------------- extended Jacobian (XYZZ) */

static inline void GF(xyzz_set_infinity)(XYZZ *p) { /* g1.go:688: (1,1,0,0) */
    CFF(set_one)(&p->x); CFF(set_one)(&p->y); CFF(set_zero)(&p->zz); CFF(set_zero)(&p->zzz);
}
static inline int GF(xyzz_is_infinity)(const XYZZ *p) { return CFF(is_zero)(&p->zz); } /* g1.go:697 */
static inline int GF(aff_is_infinity)(const AFF *a) { return CFF(is_zero)(&a->x) && CFF(is_zero)(&a->y); } /* g1.go:178 */

/* doubleMixed / doubleNegMixed (g1.go:962 / :933): p = [2](+-a), a affine (dbl-2008-s-1 with ZZ=ZZZ=1) */
static inline void GF(xyzz_double_mixed)(XYZZ *p, const AFF *a, int negate) {
    CFT U, V, W, S, XX, M, S2, L;
    CFF(dbl)(&U, &a->y);
    if (negate) CFF(neg)(&U, &U);
    CFF(sqr)(&V, &U);
    CFF(mul)(&W, &U, &V);
    CFF(mul)(&S, &a->x, &V);
    CFF(sqr)(&XX, &a->x);
    CFF(dbl)(&M, &XX);
    CFF(add)(&M, &M, &XX);
    CFF(dbl)(&S2, &S);
    CFF(mul)(&L, &W, &a->y);
    CFF(sqr)(&p->x, &M);
    CFF(sub)(&p->x, &p->x, &S2);
    CFF(sub)(&p->y, &S, &p->x);
    CFF(mul)(&p->y, &p->y, &M);
    if (negate) CFF(add)(&p->y, &p->y, &L); else CFF(sub)(&p->y, &p->y, &L);
    p->zz = V;
    p->zzz = W;
}

/* addMixed / subMixed (g1.go:822 / :878): p += (+-a), madd-2008-s */
static inline void GF(xyzz_add_mixed)(XYZZ *p, const AFF *a, int negate) {
    if (GF(aff_is_infinity)(a)) return;
    if (CFF(is_zero)(&p->zz)) {
        p->x = a->x;
        if (negate) CFF(neg)(&p->y, &a->y); else p->y = a->y;
        CFF(set_one)(&p->zz);
        CFF(set_one)(&p->zzz);
        return;
    }
    CFT P, R;
    CFF(mul)(&P, &a->x, &p->zz);
    CFF(sub)(&P, &P, &p->x);
    CFF(mul)(&R, &a->y, &p->zzz);
    if (negate) CFF(neg)(&R, &R);
    CFF(sub)(&R, &R, &p->y);
    if (CFF(is_zero)(&P)) {
        if (CFF(is_zero)(&R)) { GF(xyzz_double_mixed)(p, a, negate); return; }
        CFF(set_zero)(&p->zz);
        CFF(set_zero)(&p->zzz);
        return;
    }
    CFT PP, PPP, Q, Q2, RR, X3, Y3;
    CFF(sqr)(&PP, &P);
    CFF(mul)(&PPP, &P, &PP);
    CFF(mul)(&Q, &p->x, &PP);
    CFF(sqr)(&RR, &R);
    CFF(sub)(&X3, &RR, &PPP);
    CFF(dbl)(&Q2, &Q);
    CFF(sub)(&p->x, &X3, &Q2);
    CFF(sub)(&Y3, &Q, &p->x);
    CFF(mul)(&Y3, &Y3, &R);
    CFF(mul)(&R, &p->y, &PPP);
    CFF(sub)(&p->y, &Y3, &R);
    CFF(mul)(&p->zz, &p->zz, &PP);
    CFF(mul)(&p->zzz, &p->zzz, &PPP);
}

/* double (g1.go:795): p = [2]q, dbl-2008-s-1, a = 0 */
static inline void GF(xyzz_double)(XYZZ *p, const XYZZ *q) {
    CFT U, V, W, S, XX, M;
    XYZZ r;
    CFF(dbl)(&U, &q->y);
    CFF(sqr)(&V, &U);
    CFF(mul)(&W, &U, &V);
    CFF(mul)(&S, &q->x, &V);
    CFF(sqr)(&XX, &q->x);
    CFF(dbl)(&M, &XX);
    CFF(add)(&M, &M, &XX);
    CFF(mul)(&U, &W, &q->y);
    CFF(sqr)(&r.x, &M);
    CFF(sub)(&r.x, &r.x, &S);
    CFF(sub)(&r.x, &r.x, &S);
    CFF(sub)(&r.y, &S, &r.x);
    CFF(mul)(&r.y, &r.y, &M);
    CFF(sub)(&r.y, &r.y, &U);
    CFF(mul)(&r.zz, &V, &q->zz);
    CFF(mul)(&r.zzz, &W, &q->zzz);
    *p = r;
}

/* add (g1.go:736): p += q, add-2008-s */
static inline void GF(xyzz_add)(XYZZ *p, const XYZZ *q) {
    if (CFF(is_zero)(&q->zz)) return;
    if (CFF(is_zero)(&p->zz)) { *p = *q; return; }
    CFT A, B, U1, U2, S1, S2;
    CFF(mul)(&U2, &q->x, &p->zz);
    CFF(mul)(&U1, &p->x, &q->zz);
    CFF(sub)(&A, &U2, &U1);
    CFF(mul)(&S2, &q->y, &p->zzz);
    CFF(mul)(&S1, &p->y, &q->zzz);
    CFF(sub)(&B, &S2, &S1);
    if (CFF(is_zero)(&A)) {
        if (CFF(is_zero)(&B)) { GF(xyzz_double)(p, q); return; }
        CFF(set_zero)(&p->zz);
        CFF(set_zero)(&p->zzz);
        return;
    }
    CFT PP, PPP, Q, V;
    CFF(sqr)(&PP, &A);
    CFF(mul)(&PPP, &A, &PP);
    CFF(mul)(&Q, &U1, &PP);
    CFF(mul)(&V, &S1, &PPP);
    CFF(sqr)(&p->x, &B);
    CFF(sub)(&p->x, &p->x, &PPP);
    CFF(sub)(&p->x, &p->x, &Q);
    CFF(sub)(&p->x, &p->x, &Q);
    CFF(sub)(&p->y, &Q, &p->x);
    CFF(mul)(&p->y, &p->y, &B);
    CFF(sub)(&p->y, &p->y, &V);
    CFF(mul)(&p->zz, &p->zz, &q->zz);
    CFF(mul)(&p->zz, &p->zz, &PP);
    CFF(mul)(&p->zzz, &p->zzz, &q->zzz);
    CFF(mul)(&p->zzz, &p->zzz, &PPP);
}

/* ---------------------------------------------------------------- Jacobian */

static inline void GF(jac_set_infinity)(JAC *p) { /* bn254.go:125-129: (1,1,0) */
    CFF(set_one)(&p->x); CFF(set_one)(&p->y); CFF(set_zero)(&p->z);
}

/* fromJacExtended (g1.go:713): infinity-safe; unsafeFromJacExtended (:726) is the same formula without the check
 * (X*ZZ^2, Y*ZZZ^2, ZZZ), so on infinity it would give Z = 0 as well. */
static inline void GF(jac_from_xyzz)(JAC *p, const XYZZ *q) {
    if (CFF(is_zero)(&q->zz)) { GF(jac_set_infinity)(p); return; }
    CFF(sqr)(&p->x, &q->zz);
    CFF(mul)(&p->x, &p->x, &q->x);
    CFF(sqr)(&p->y, &q->zzz);
    CFF(mul)(&p->y, &p->y, &q->y);
    p->z = q->zzz;
}

static inline void GF(jac_double_assign)(JAC *p) { /* g1.go:396 dbl-2007-bl */
    CFT XX, YY, YYYY, ZZ, S, M, T;
    CFF(sqr)(&XX, &p->x);
    CFF(sqr)(&YY, &p->y);
    CFF(sqr)(&YYYY, &YY);
    CFF(sqr)(&ZZ, &p->z);
    CFF(add)(&S, &p->x, &YY);
    CFF(sqr)(&S, &S);
    CFF(sub)(&S, &S, &XX);
    CFF(sub)(&S, &S, &YYYY);
    CFF(dbl)(&S, &S);
    CFF(dbl)(&M, &XX);
    CFF(add)(&M, &M, &XX);
    CFF(add)(&p->z, &p->z, &p->y);
    CFF(sqr)(&p->z, &p->z);
    CFF(sub)(&p->z, &p->z, &YY);
    CFF(sub)(&p->z, &p->z, &ZZ);
    CFF(sqr)(&T, &M);
    p->x = T;
    CFF(dbl)(&T, &S);
    CFF(sub)(&p->x, &p->x, &T);
    CFF(sub)(&p->y, &S, &p->x);
    CFF(mul)(&p->y, &p->y, &M);
    CFF(dbl)(&YYYY, &YYYY); CFF(dbl)(&YYYY, &YYYY); CFF(dbl)(&YYYY, &YYYY);
    CFF(sub)(&p->y, &p->y, &YYYY);
}

static inline void GF(jac_add_assign)(JAC *p, const JAC *q) { /* g1.go:243 add-2007-bl */
    if (CFF(is_zero)(&p->z)) { *p = *q; return; }
    if (CFF(is_zero)(&q->z)) return;
    CFT Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, r, V;
    CFF(sqr)(&Z1Z1, &q->z);
    CFF(sqr)(&Z2Z2, &p->z);
    CFF(mul)(&U1, &q->x, &Z2Z2);
    CFF(mul)(&U2, &p->x, &Z1Z1);
    CFF(mul)(&S1, &q->y, &p->z); CFF(mul)(&S1, &S1, &Z2Z2);
    CFF(mul)(&S2, &p->y, &q->z); CFF(mul)(&S2, &S2, &Z1Z1);
    if (CFF(equal)(&U1, &U2) && CFF(equal)(&S1, &S2)) { GF(jac_double_assign)(p); return; }
    CFF(sub)(&H, &U2, &U1);
    CFF(dbl)(&I, &H); CFF(sqr)(&I, &I);
    CFF(mul)(&J, &H, &I);
    CFF(sub)(&r, &S2, &S1); CFF(dbl)(&r, &r);
    CFF(mul)(&V, &U1, &I);
    CFF(sqr)(&p->x, &r);
    CFF(sub)(&p->x, &p->x, &J);
    CFF(sub)(&p->x, &p->x, &V);
    CFF(sub)(&p->x, &p->x, &V);
    CFF(sub)(&p->y, &V, &p->x);
    CFF(mul)(&p->y, &p->y, &r);
    CFF(mul)(&S1, &S1, &J); CFF(dbl)(&S1, &S1);
    CFF(sub)(&p->y, &p->y, &S1);
    CFF(add)(&p->z, &p->z, &q->z);
    CFF(sqr)(&p->z, &p->z);
    CFF(sub)(&p->z, &p->z, &Z1Z1);
    CFF(sub)(&p->z, &p->z, &Z2Z2);
    CFF(mul)(&p->z, &p->z, &H);
}

static inline void GF(aff_from_jac)(AFF *p, const JAC *q) { /* g1.go:150 FromJacobian */
    if (CFF(is_zero)(&q->z)) { CFF(set_zero)(&p->x); CFF(set_zero)(&p->y); return; }
    CFT a, b;
    CFF(inv)(&a, &q->z);
    CFF(sqr)(&b, &a);
    CFF(mul)(&p->x, &q->x, &b);
    CFF(mul)(&p->y, &q->y, &b);
    CFF(mul)(&p->y, &p->y, &a);
}

/* ---------------------------------------------------------------- scalar partition */

static inline unsigned GF(nb_chunks)(unsigned c) { return (SF_BITS + c - 1) / c; } /* multiexp.go:681 */

/* multiexp.go:709-803. digits is window-major: digits[chunk*n + i]; encoding 0 = skip, d>0 -> 2d,
 * d<0 -> 2(-d-1)+1.  `bits` of a scalar = its non-Montgomery value (fr.Element.Bits, fr/element.go:855). */
static void GF(partition_range)(const SFT *scalars, size_t n, unsigned c, uint16_t *digits, size_t start, size_t end) {
    const unsigned nb = GF(nb_chunks)(c);
    const int max = (1 << (c - 1)) - 1;
    for (size_t i = start; i < end; ++i) {
        if (SFF(is_zero)(&scalars[i])) continue; /* digits pre-zeroed */
        SFT s = scalars[i];
        SFF(from_mont)(&s);
        int carry = 0;
        for (unsigned chunk = 0; chunk < nb; ++chunk) {
            /* selector (multiexp.go:728-741): c bits starting at bit chunk*c, possibly spanning two words,
             * the top window simply runs out of words */
            unsigned jc = chunk * c, idx = jc / 64, sh = jc % 64;
            uint64_t w = s.l[idx] >> sh;
            if (sh + c > 64 && idx + 1 < SF_N) w |= s.l[idx + 1] << (64 - sh);
            int digit = carry + (int)(w & ((1ull << c) - 1));
            if (chunk < nb - 1) {
                carry = 0;
                if (digit > max) { digit -= 1 << c; carry = 1; }
                if (digit == 0) continue;
                uint16_t bits = digit > 0 ? (uint16_t)(digit << 1) : (uint16_t)((((-digit) - 1) << 1) + 1);
                digits[(size_t)chunk * n + i] = bits;
            } else {
                digits[(size_t)chunk * n + i] = (uint16_t)(digit << 1); /* top window: no borrow (:788-800) */
            }
        }
    }
}

/* ---------------------------------------------------------------- one window */

/* multiexp_jacobian.go:8-61 */
static void GF(process_chunk)(unsigned c_buckets, const AFF *points, const uint16_t *digits, size_t n, XYZZ *buckets, XYZZ *out) {
    const size_t nbuckets = (size_t)1 << (c_buckets - 1);
    for (size_t k = 0; k < nbuckets; ++k) GF(xyzz_set_infinity)(&buckets[k]);
    for (size_t i = 0; i < n; ++i) {
        uint16_t d = digits[i];
        if (d == 0) continue;
        if ((d & 1) == 0) GF(xyzz_add_mixed)(&buckets[(d >> 1) - 1], &points[i], 0);
        else GF(xyzz_add_mixed)(&buckets[d >> 1], &points[i], 1);
    }
    XYZZ running, total;
    GF(xyzz_set_infinity)(&running);
    GF(xyzz_set_infinity)(&total);
    for (size_t k = nbuckets; k-- > 0;) {
        if (!GF(xyzz_is_infinity)(&buckets[k])) GF(xyzz_add)(&running, &buckets[k]);
        GF(xyzz_add)(&total, &running);
    }
    *out = total;
}

/* ---------------------------------------------------------------- one window, batch-affine buckets */

#ifndef ORACLE_BATCH_AFFINE_DEFINED
#define ORACLE_BATCH_AFFINE_DEFINED
static int oracle_batch_affine_on = 1;
/* batch sizes of the generated code (multiexp_affine.go:296-340): index = c */
static const unsigned oracle_batch_size[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 80, 150, 200, 350, 400, 500, 640};
#endif

/* batchAddG1Affine (g1.go:1122-1182): R[j] += P[j] for cnt independent pairs with ONE field inversion */
static void GF(batch_add_affine)(AFF **R, const AFF *P, unsigned cnt, CFT *lambda, CFT *lambdain) {
    if (cnt == 0) return;
    for (unsigned j = 0; j < cnt; ++j) CFF(sub)(&lambdain[j], &P[j].x, &R[j]->x);
    CFT acc;
    CFF(set_one)(&lambda[0]);
    acc = lambdain[0];
    for (unsigned i = 1; i < cnt; ++i) {
        lambda[i] = acc;
        CFF(mul)(&acc, &acc, &lambdain[i]);
    }
    CFF(inv)(&acc, &acc);
    for (unsigned i = cnt - 1; i > 0; --i) {
        CFF(mul)(&lambda[i], &lambda[i], &acc);
        CFF(mul)(&acc, &acc, &lambdain[i]);
    }
    lambda[0] = acc;
    for (unsigned j = 0; j < cnt; ++j) {
        CFT t;
        AFF Q;
        CFF(sub)(&t, &P[j].y, &R[j]->y);
        CFF(mul)(&lambda[j], &lambda[j], &t);      /* lambda = (y2 - y1) / (x2 - x1) */
        CFF(sqr)(&Q.x, &lambda[j]);
        CFF(sub)(&Q.x, &Q.x, &R[j]->x);
        CFF(sub)(&Q.x, &Q.x, &P[j].x);             /* x3 = lambda^2 - x1 - x2 */
        CFF(sub)(&t, &R[j]->x, &Q.x);
        CFF(mul)(&Q.y, &lambda[j], &t);
        CFF(sub)(&Q.y, &Q.y, &R[j]->y);            /* y3 = lambda (x1 - x3) - y1 */
        *R[j] = Q;
    }
}

typedef struct { uint16_t bucket; AFF point; } GF(bop_t);

/* processChunkG1BatchAffine (multiexp_affine.go:24-231). buckets_je: 2^(c_buckets-1) XYZZ (the caller's scratch, also
 * used by the Jacobian variant); the affine buckets, batch and queue are allocated here. */
static void GF(process_chunk_batch_affine)(unsigned c_buckets, unsigned batch, const AFF *points, const uint16_t *digits,
                                           size_t n, XYZZ *buckets_je, XYZZ *out) {
    const size_t nbuckets = (size_t)1 << (c_buckets - 1);
    AFF *buckets = (AFF *)calloc(nbuckets, sizeof(AFF));         /* (0,0) = infinity */
    uint8_t *in_batch = (uint8_t *)calloc(nbuckets, 1);          /* bitSet: bucket already part of the current batch */
    AFF **R = (AFF **)malloc(sizeof(AFF *) * batch);
    AFF *P = (AFF *)malloc(sizeof(AFF) * batch);
    CFT *lambda = (CFT *)malloc(sizeof(CFT) * batch), *lambdain = (CFT *)malloc(sizeof(CFT) * batch);
    GF(bop_t) *queue = (GF(bop_t) *)malloc(sizeof(GF(bop_t)) * batch);
    unsigned cnt = 0, qid = 0;
    for (size_t k = 0; k < nbuckets; ++k) GF(xyzz_set_infinity)(&buckets_je[k]);

#define ORACLE_EXECUTE_AND_RESET()                                                   \
    do {                                                                             \
        GF(batch_add_affine)(R, P, cnt, lambda, lambdain);                           \
        for (unsigned q_ = 0; q_ < cnt; ++q_) in_batch[R[q_] - buckets] = 0;          \
        cnt = 0;                                                                     \
    } while (0)

    for (size_t i = 0; i < n; ++i) {
        const uint16_t digit = digits[i];
        if (digit == 0 || GF(aff_is_infinity)(&points[i])) continue;
        uint16_t bid = (uint16_t)(digit >> 1);
        const int is_add = (digit & 1) == 0;
        if (is_add) bid -= 1;
        AFF pt = points[i];
        if (!is_add) CFF(neg)(&pt.y, &pt.y);
        if (in_batch[bid]) {  /* conflict: queue it (multiexp_affine.go:181-197) */
            queue[qid].bucket = bid;
            queue[qid].point = pt;
            ++qid;
            if (qid == batch - 1) {  /* queue full: flush into the extended-Jacobian buckets */
                for (unsigned q = 0; q < qid; ++q) GF(xyzz_add_mixed)(&buckets_je[queue[q].bucket], &queue[q].point, 0);
                qid = 0;
            }
            continue;
        }
        /* add(): special cases first (multiexp_affine.go:113-149); pt already carries the sign */
        AFF *BK = &buckets[bid];
        if (GF(aff_is_infinity)(BK)) { *BK = pt; continue; }
        if (CFF(equal)(&BK->x, &pt.x)) {
            if (CFF(equal)(&BK->y, &pt.y)) GF(xyzz_add_mixed)(&buckets_je[bid], &pt, 0);  /* P + P: rare, other bucket set */
            else { CFF(set_zero)(&BK->x); CFF(set_zero)(&BK->y); }                          /* P - P */
            continue;
        }
        in_batch[bid] = 1;
        R[cnt] = BK;
        P[cnt] = pt;
        ++cnt;
        if (cnt == batch) {
            ORACLE_EXECUTE_AND_RESET();
            /* processTopQueue (multiexp_affine.go:158-169) */
            while (qid > 0) {
                GF(bop_t) *op = &queue[qid - 1];
                if (in_batch[op->bucket]) break;
                AFF *B2 = &buckets[op->bucket];
                if (GF(aff_is_infinity)(B2)) *B2 = op->point;
                else if (CFF(equal)(&B2->x, &op->point.x)) {
                    if (CFF(equal)(&B2->y, &op->point.y)) GF(xyzz_add_mixed)(&buckets_je[op->bucket], &op->point, 0);
                    else { CFF(set_zero)(&B2->x); CFF(set_zero)(&B2->y); }
                } else {
                    in_batch[op->bucket] = 1;
                    R[cnt] = B2;
                    P[cnt] = op->point;
                    ++cnt;
                }
                --qid;
            }
        }
    }
    ORACLE_EXECUTE_AND_RESET();
    for (unsigned q = 0; q < qid; ++q) GF(xyzz_add_mixed)(&buckets_je[queue[q].bucket], &queue[q].point, 0);
#undef ORACLE_EXECUTE_AND_RESET
    /* total = bucket[0] + 2 bucket[1] + ... (multiexp_affine.go:207-218) */
    XYZZ running, total;
    GF(xyzz_set_infinity)(&running);
    GF(xyzz_set_infinity)(&total);
    for (size_t k = nbuckets; k-- > 0;) {
        GF(xyzz_add_mixed)(&running, &buckets[k], 0);
        if (!GF(xyzz_is_infinity)(&buckets_je[k])) GF(xyzz_add)(&running, &buckets_je[k]);
        GF(xyzz_add)(&total, &running);
    }
    *out = total;
    free(buckets); free(in_batch); free(R); free(P); free(lambda); free(lambdain); free(queue);
}

/* getChunkProcessorG1 (multiexp.go:213-299): batch-affine buckets for c >= 10 when at least batchSize distinct
 * buckets of the window are hit (chunkStat.nbBucketFilled, multiexp.go:811-838), extended-Jacobian otherwise. */
static void GF(process_chunk_auto)(unsigned c, unsigned c_buckets, const AFF *points, const uint16_t *digits, size_t n,
                                   XYZZ *buckets, XYZZ *out) {
    const unsigned batch = (oracle_batch_affine_on && c >= 10 && c <= 16) ? oracle_batch_size[c] : 0;
    if (batch) {
        const size_t nbuckets = (size_t)1 << (c_buckets - 1);
        uint8_t *hit = (uint8_t *)calloc(nbuckets, 1);
        size_t nz = 0;
        for (size_t i = 0; i < n && nz < batch; ++i) {
            const uint16_t d = digits[i];
            if (d == 0) continue;
            const size_t b = (d & 1) ? (size_t)(d >> 1) : (size_t)(d >> 1) - 1;
            if (!hit[b]) { hit[b] = 1; ++nz; }
        }
        free(hit);
        if (nz >= batch) {
            GF(process_chunk_batch_affine)(c_buckets, batch, points, digits, n, buckets, out);
            return;
        }
    }
    GF(process_chunk)(c_buckets, points, digits, n, buckets, out);
}

typedef struct {
    unsigned c;        /* digit window width */
    unsigned c_alloc;  /* bucket arrays hold 2^(c_alloc-1) entries: max(c, lastC(c)) */
    size_t n;
    const AFF *points;
    const SFT *scalars;
    uint16_t *digits;
    XYZZ *totals;
    unsigned nb;
    int next;          /* next window to hand out (top first, like the reference's spawn order) */
    size_t part_next;  /* next scalar block for the partition phase */
    pthread_mutex_t mu;
} GF(job_t);

static void *GF(partition_worker)(void *arg) {
    GF(job_t) *job = (GF(job_t) *)arg;
    const size_t blk = 4096;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        size_t s = job->part_next;
        job->part_next += blk;
        pthread_mutex_unlock(&job->mu);
        if (s >= job->n) break;
        size_t e = s + blk < job->n ? s + blk : job->n;
        GF(partition_range)(job->scalars, job->n, job->c, job->digits, s, e);
    }
    return NULL;
}

static void *GF(chunk_worker)(void *arg) {
    GF(job_t) *job = (GF(job_t) *)arg;
    XYZZ *buckets = (XYZZ *)malloc(sizeof(XYZZ) << (job->c_alloc - 1));
    for (;;) {
        pthread_mutex_lock(&job->mu);
        int j = job->next--;
        pthread_mutex_unlock(&job->mu);
        if (j < 0) break;
        /* the reference sizes the top window's bucket array from lastC(c) (multiexp.go:182-184); every worker
         * here gets 2^(max(c,lastC)-1) buckets, unused high buckets stay at infinity and cost only adds of
         * infinity in the reduction (no effect on the value). */
        GF(process_chunk_auto)(job->c, job->c_alloc, job->points, job->digits + (size_t)j * job->n, job->n, buckets, &job->totals[j]);
    }
    free(buckets);
    return NULL;
}

/* _innerMsmG1 (multiexp.go:148-209) + msmReduceChunk (:302-315). Returns the Jacobian result. */
static void GF(inner_msm)(JAC *out, unsigned c, const AFF *points, const SFT *scalars, size_t n, int nthreads) {
    const unsigned nb = GF(nb_chunks)(c);
    GF(job_t) job;
    job.c = c; job.n = n; job.points = points; job.scalars = scalars; job.nb = nb;
    job.digits = (uint16_t *)calloc((size_t)nb * (n ? n : 1), sizeof(uint16_t));
    job.totals = (XYZZ *)malloc(sizeof(XYZZ) * nb);
    job.next = (int)nb - 1;
    job.part_next = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    /* lastC (multiexp.go:690): the top window holds bits_top = c - avail bits plus a carry, i.e. a digit of at
     * most 2^(lastC-1) with lastC = c + 1 - avail, which needs 2^(lastC-1) buckets. */
    unsigned avail = nb * c - SF_BITS;
    unsigned lastc = c + 1 - avail;
    job.c_alloc = lastc > c ? lastc : c;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    if (nthreads == 1) GF(partition_worker)(&job);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, GF(partition_worker), &job);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    int nw = nthreads < (int)nb ? nthreads : (int)nb;
    if (nw == 1) GF(chunk_worker)(&job);
    else {
        for (int t = 0; t < nw; ++t) pthread_create(&th[t], NULL, GF(chunk_worker), &job);
        for (int t = 0; t < nw; ++t) pthread_join(th[t], NULL);
    }
    free(th);
    /* msmReduceChunk: Horner from the top window down */
    XYZZ acc = job.totals[nb - 1];
    for (int j = (int)nb - 2; j >= 0; --j) {
        for (unsigned l = 0; l < c; ++l) GF(xyzz_double)(&acc, &acc);
        GF(xyzz_add)(&acc, &job.totals[j]);
    }
    GF(jac_from_xyzz)(out, &acc);
    pthread_mutex_destroy(&job.mu);
    free(job.digits);
    free(job.totals);
}

/* bestC (multiexp.go:75-93) */
static unsigned GF(best_c)(size_t n) {
    static const unsigned cs[] = G_CS;
    double min = INFINITY;
    unsigned C = 0;
    for (size_t k = 0; k < sizeof cs / sizeof cs[0]; ++k) {
        unsigned c = cs[k];
        double cc = (double)(SF_BITS + 1) * (double)(n + ((size_t)1 << c));
        double cost = cc / (double)c;
        if (cost < min) { min = cost; C = c; }
    }
    return C;
}

static long GF(cost_function)(long nb_tasks, long nb_cpus, long cost_per_task) { /* multiexp.go:103-116 */
    long total = nb_tasks;
    while (nb_tasks >= nb_cpus) { nb_tasks -= nb_cpus; total += cost_per_task; }
    if (nb_tasks > 0) total += cost_per_task;
    return total;
}

/* (*G1Jac).MultiExp (multiexp.go:32-146). num_cpu stands in for runtime.NumCPU(); nthreads = worker threads
 * actually used.  Returns 0 ok, 1 length mismatch, 2 bad config -- the two reference errors (:61-71).
 *
 * The reference recursion (split in halves while that lowers its cost model, each half in its own goroutine, every
 * window of every leaf in its own goroutine) is restated as: plan the same leaves with the same cost model, run all
 * (leaf, window) tasks of all leaves on one pool of nthreads workers, then combine exactly like the recursion does
 * (p = MultiExp(upper half); p.AddAssign(MultiExp(lower half)), multiexp.go:128-139). */
typedef struct {
    size_t off, n;
    unsigned c, c_alloc, nb;
    uint16_t *digits;
    XYZZ *totals;
    JAC result;
} GF(leaf_t);

typedef struct {
    GF(leaf_t) *leaves;
    size_t nleaves, cap;
} GF(plan_t);

static void GF(plan_rec)(GF(plan_t) *plan, size_t off, size_t n, int nb_tasks) {
    unsigned C = GF(best_c)(n);
    long nbc = GF(nb_chunks)(C);
    long pre = GF(cost_function)(nbc, nb_tasks, (long)(n + ((size_t)1 << C)));
    unsigned c2 = GF(best_c)(n / 2);
    long post = GF(cost_function)(2 * (long)GF(nb_chunks)(c2), nb_tasks, (long)(n / 2 + ((size_t)1 << c2)));
    if (post < pre) {
        int half_tasks = (nb_tasks + 1) / 2; /* ceil(nbTasks/2) */
        GF(plan_rec)(plan, off, n / 2, half_tasks);               /* _p: points[:n/2] */
        GF(plan_rec)(plan, off + n / 2, n - n / 2, half_tasks);   /* p:  points[n/2:] */
        return;
    }
    if (plan->nleaves == plan->cap) {
        plan->cap = plan->cap ? plan->cap * 2 : 16;
        plan->leaves = (GF(leaf_t) *)realloc(plan->leaves, plan->cap * sizeof(GF(leaf_t)));
    }
    GF(leaf_t) *lf = &plan->leaves[plan->nleaves++];
    lf->off = off; lf->n = n; lf->c = C; lf->nb = GF(nb_chunks)(C);
    unsigned avail = lf->nb * C - SF_BITS, lastc = C + 1 - avail;
    lf->c_alloc = lastc > C ? lastc : C;
    lf->digits = NULL; lf->totals = NULL;
}

/* same recursion again, consuming leaf results in plan order */
static void GF(combine_rec)(GF(plan_t) *plan, size_t *next, size_t n, int nb_tasks, JAC *out) {
    unsigned C = GF(best_c)(n);
    long nbc = GF(nb_chunks)(C);
    long pre = GF(cost_function)(nbc, nb_tasks, (long)(n + ((size_t)1 << C)));
    unsigned c2 = GF(best_c)(n / 2);
    long post = GF(cost_function)(2 * (long)GF(nb_chunks)(c2), nb_tasks, (long)(n / 2 + ((size_t)1 << c2)));
    if (post < pre) {
        int half_tasks = (nb_tasks + 1) / 2;
        JAC lo;
        GF(combine_rec)(plan, next, n / 2, half_tasks, &lo);
        GF(combine_rec)(plan, next, n - n / 2, half_tasks, out);
        GF(jac_add_assign)(out, &lo);
        return;
    }
    *out = plan->leaves[(*next)++].result;
}

typedef struct {
    GF(plan_t) *plan;
    const AFF *points;
    const SFT *scalars;
    size_t next_part;   /* partition phase: (leaf, block) cursor flattened over leaves */
    size_t next_task;   /* chunk phase: flattened (leaf, window) cursor */
    size_t ntasks;
    unsigned max_c_alloc;
    pthread_mutex_t mu;
} GF(pool_t);

static void *GF(pool_partition)(void *arg) {
    GF(pool_t) *pool = (GF(pool_t) *)arg;
    const size_t blk = 4096;
    for (;;) {
        pthread_mutex_lock(&pool->mu);
        size_t cur = pool->next_part;
        pool->next_part += blk;
        pthread_mutex_unlock(&pool->mu);
        /* cur indexes the concatenation of all leaves (they tile [0, n_total) in order) */
        GF(plan_t) *plan = pool->plan;
        GF(leaf_t) *last = &plan->leaves[plan->nleaves - 1];
        if (cur >= last->off + last->n) break;
        for (size_t l = 0; l < plan->nleaves; ++l) {
            GF(leaf_t) *lf = &plan->leaves[l];
            size_t s = cur > lf->off ? cur : lf->off;
            size_t e = cur + blk < lf->off + lf->n ? cur + blk : lf->off + lf->n;
            if (s < e) GF(partition_range)(pool->scalars + lf->off, lf->n, lf->c, lf->digits, s - lf->off, e - lf->off);
        }
    }
    return NULL;
}

static void *GF(pool_chunks)(void *arg) {
    GF(pool_t) *pool = (GF(pool_t) *)arg;
    XYZZ *buckets = (XYZZ *)malloc(sizeof(XYZZ) << (pool->max_c_alloc - 1));
    for (;;) {
        pthread_mutex_lock(&pool->mu);
        size_t t = pool->next_task++;
        pthread_mutex_unlock(&pool->mu);
        if (t >= pool->ntasks) break;
        /* locate (leaf, window): windows handed out top first within each leaf */
        GF(plan_t) *plan = pool->plan;
        size_t l = 0;
        while (t >= plan->leaves[l].nb) { t -= plan->leaves[l].nb; ++l; }
        GF(leaf_t) *lf = &plan->leaves[l];
        unsigned j = lf->nb - 1 - (unsigned)t;
        GF(process_chunk_auto)(lf->c, lf->c_alloc, pool->points + lf->off, lf->digits + (size_t)j * lf->n, lf->n, buckets, &lf->totals[j]);
    }
    free(buckets);
    return NULL;
}

static int GF(multiexp)(JAC *out, const AFF *points, size_t n_points, const SFT *scalars, size_t n_scalars,
                        int nb_tasks, int num_cpu, int nthreads) {
    if (n_points != n_scalars) return 1;
    if (nb_tasks <= 0) nb_tasks = num_cpu * 2;
    else if (nb_tasks > 1024) return 2;
    if (nthreads < 1) nthreads = 1;
    GF(plan_t) plan = {NULL, 0, 0};
    GF(plan_rec)(&plan, 0, n_points, nb_tasks);
    GF(pool_t) pool;
    pool.plan = &plan; pool.points = points; pool.scalars = scalars;
    pool.next_part = 0; pool.next_task = 0; pool.ntasks = 0; pool.max_c_alloc = 2;
    pthread_mutex_init(&pool.mu, NULL);
    for (size_t l = 0; l < plan.nleaves; ++l) {
        GF(leaf_t) *lf = &plan.leaves[l];
        lf->digits = (uint16_t *)calloc((size_t)lf->nb * (lf->n ? lf->n : 1), sizeof(uint16_t));
        lf->totals = (XYZZ *)malloc(sizeof(XYZZ) * lf->nb);
        pool.ntasks += lf->nb;
        if (lf->c_alloc > pool.max_c_alloc) pool.max_c_alloc = lf->c_alloc;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    if (n_points) {
        if (nthreads == 1) GF(pool_partition)(&pool);
        else {
            for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, GF(pool_partition), &pool);
            for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        }
    }
    int nw = (size_t)nthreads < pool.ntasks ? nthreads : (int)pool.ntasks;
    if (nw <= 1) GF(pool_chunks)(&pool);
    else {
        for (int t = 0; t < nw; ++t) pthread_create(&th[t], NULL, GF(pool_chunks), &pool);
        for (int t = 0; t < nw; ++t) pthread_join(th[t], NULL);
    }
    free(th);
    for (size_t l = 0; l < plan.nleaves; ++l) { /* msmReduceChunk per leaf */
        GF(leaf_t) *lf = &plan.leaves[l];
        XYZZ acc = lf->totals[lf->nb - 1];
        for (int j = (int)lf->nb - 2; j >= 0; --j) {
            for (unsigned k = 0; k < lf->c; ++k) GF(xyzz_double)(&acc, &acc);
            GF(xyzz_add)(&acc, &lf->totals[j]);
        }
        GF(jac_from_xyzz)(&lf->result, &acc);
        free(lf->digits);
        free(lf->totals);
    }
    size_t next = 0;
    GF(combine_rec)(&plan, &next, n_points, nb_tasks, out);
    pthread_mutex_destroy(&pool.mu);
    free(plan.leaves);
    return 0;
}

/* ---------------------------------------------------------------- helpers for tests / input generation */

/* [k]a by left-to-right double-and-add over plain (non-Montgomery) little-endian scalar limbs. */
static void GF(scalar_mul)(XYZZ *out, const AFF *a, const uint64_t *k, int klimbs) {
    XYZZ acc;
    GF(xyzz_set_infinity)(&acc);
    for (int i = klimbs * 64 - 1; i >= 0; --i) {
        GF(xyzz_double)(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) GF(xyzz_add_mixed)(&acc, a, 0);
    }
    *out = acc;
}

static void GF(aff_from_xyzz)(AFF *p, const XYZZ *q) { /* via Jacobian so that the value path == FromJacobian */
    JAC j;
    GF(jac_from_xyzz)(&j, q);
    GF(aff_from_jac)(p, &j);
}

/* Normalise count XYZZ points to affine with one inversion (Montgomery's trick): x = X/ZZ, y = Y/ZZZ. */
static void GF(batch_xyzz_to_aff)(AFF *out, const XYZZ *in, size_t count, CFT *scratch) {
    /* scratch: count elements, prefix products of zzz; 1/zz is derived as zz^2 * (1/zzz)^2. */
    CFT acc;
    CFF(set_one)(&acc);
    for (size_t i = 0; i < count; ++i) {
        scratch[i] = acc;
        if (!CFF(is_zero)(&in[i].zzz)) CFF(mul)(&acc, &acc, &in[i].zzz);
    }
    CFT inv;
    CFF(inv)(&inv, &acc);
    for (size_t i = count; i-- > 0;) {
        if (CFF(is_zero)(&in[i].zzz)) { CFF(set_zero)(&out[i].x); CFF(set_zero)(&out[i].y); continue; }
        CFT zi, zi2, izz;
        CFF(mul)(&zi, &inv, &scratch[i]);          /* 1/zzz_i */
        CFF(mul)(&inv, &inv, &in[i].zzz);
        CFF(sqr)(&zi2, &zi);
        CFF(mul)(&izz, &zi2, &in[i].zz);
        CFF(mul)(&izz, &izz, &in[i].zz);           /* zz^2/zzz^2 = 1/zz  (zz^3 = zzz^2) */
        CFF(mul)(&out[i].x, &in[i].x, &izz);
        CFF(mul)(&out[i].y, &in[i].y, &zi);
    }
}

typedef struct {
    const AFF *base, *step;
    const uint64_t *k0, *k1;
    int klimbs;
    size_t start, end;
    AFF *out;
} GF(gen_t);

/* out[i] = [k0 + i*k1] g for i in [start,end): start point by scalar mul, then repeated mixed addition of
 * step = [k1]g and block-wise batch normalisation (same pattern as multiexp_test.go:40-46, which walks i*G). */
static void *GF(gen_worker)(void *arg) {
    GF(gen_t) *g = (GF(gen_t) *)arg;
    enum { BLK = 1024 };
    XYZZ *blk = (XYZZ *)malloc(sizeof(XYZZ) * BLK);
    CFT *scr = (CFT *)malloc(sizeof(CFT) * BLK);
    /* cur = [k0]base + [start]step */
    XYZZ cur, t;
    GF(scalar_mul)(&cur, g->base, g->k0, g->klimbs);
    uint64_t s64 = (uint64_t)g->start;
    GF(scalar_mul)(&t, g->step, &s64, 1);
    GF(xyzz_add)(&cur, &t);
    size_t i = g->start;
    while (i < g->end) {
        size_t cnt = g->end - i < BLK ? g->end - i : BLK;
        for (size_t k = 0; k < cnt; ++k) {
            blk[k] = cur;
            GF(xyzz_add_mixed)(&cur, g->step, 0);
        }
        GF(batch_xyzz_to_aff)(g->out + i, blk, cnt, scr);
        i += cnt;
    }
    free(blk);
    free(scr);
    return NULL;
}

#undef GF
#undef CFT
#undef CFF
#undef SFT
#undef SFF
#undef AFF
#undef JAC
#undef XYZZ
