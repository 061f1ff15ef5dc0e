/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 *
 * CPU restatement (plain C, one file + three templates) of gnark-crypto's ecc/<curve>.MultiExp path for
 * BN254 (G1, G2), BLS12-381 (G1, G2) and BW6-761 (G1, G2).  See fp_tmpl.h / e2_tmpl.h / curve_tmpl.h for the
 * reference file:line each function follows.
 *
 * PINNING STATUS: the Go reference cannot be executed in this image (no Go toolchain, no network) and its
 * tests hold no stored MSM outputs.  This oracle is pinned by (tests/test_oracle_pinning.py):
 *   - field ops vs Python big-int arithmetic (the reference's own field tests compare against math/big),
 *   - the reference's algebraic MSM identities (multiexp_test.go:54-60,95-126,186-216): MSM({iG},{i*m}) =
 *     m*n(n+1)(2n+1)/6 * G for every window size c, all-infinity / all-zero cases, duplicated pairs,
 *   - the RFC 9380 known-answer points P = Q0 + Q1 stored in the reference's hash_vectors_test.go files
 *     (BN254 G1/G2, BLS12-381 G1/G2), fed through the oracle's group law and a 2-term MSM,
 *   - an independent pure-Python affine double-and-add MSM (oracle/pyref.py).
 */
#include "oracle_params.h"

/* ------------------------------------------------------------------ fields */
#define FP bn254_fp
#define FP_N BN254_FP_LIMBS
#define FP_Q bn254_fp_q
#define FP_QINVNEG BN254_FP_QINVNEG
#define FP_ONE bn254_fp_one
#define FP_RSQ bn254_fp_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define FP bn254_fr
#define FP_N BN254_FR_LIMBS
#define FP_Q bn254_fr_q
#define FP_QINVNEG BN254_FR_QINVNEG
#define FP_ONE bn254_fr_one
#define FP_RSQ bn254_fr_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define FP bls12_381_fp
#define FP_N BLS12_381_FP_LIMBS
#define FP_Q bls12_381_fp_q
#define FP_QINVNEG BLS12_381_FP_QINVNEG
#define FP_ONE bls12_381_fp_one
#define FP_RSQ bls12_381_fp_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define FP bls12_381_fr
#define FP_N BLS12_381_FR_LIMBS
#define FP_Q bls12_381_fr_q
#define FP_QINVNEG BLS12_381_FR_QINVNEG
#define FP_ONE bls12_381_fr_one
#define FP_RSQ bls12_381_fr_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define FP bw6_761_fp
#define FP_N BW6_761_FP_LIMBS
#define FP_Q bw6_761_fp_q
#define FP_QINVNEG BW6_761_FP_QINVNEG
#define FP_ONE bw6_761_fp_one
#define FP_RSQ bw6_761_fp_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define FP bw6_761_fr
#define FP_N BW6_761_FR_LIMBS
#define FP_Q bw6_761_fr_q
#define FP_QINVNEG BW6_761_FR_QINVNEG
#define FP_ONE bw6_761_fr_one
#define FP_RSQ bw6_761_fr_rsquare
#include "fp_tmpl.h"
#undef FP
#undef FP_N
#undef FP_Q
#undef FP_QINVNEG
#undef FP_ONE
#undef FP_RSQ

#define E2 bn254_e2
#define BF bn254_fp
#include "e2_tmpl.h"
#undef E2
#undef BF

#define E2 bls12_381_e2
#define BF bls12_381_fp
#include "e2_tmpl.h"
#undef E2
#undef BF

/* ------------------------------------------------------------------ groups */
#define CS_4_16 {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}
#define CS_BW6 {4, 5, 8, 10, 16}

#define G bn254_g1
#define CF bn254_fp
#define SF bn254_fr
#define SF_N BN254_FR_LIMBS
#define SF_BITS BN254_FR_BITS
#define G_CS CS_4_16
#include "curve_tmpl.h"
#undef G
#undef CF

#define G bn254_g2
#define CF bn254_e2
#include "curve_tmpl.h"
#undef G
#undef CF
#undef SF
#undef SF_N
#undef SF_BITS

#define G bls12_381_g1
#define CF bls12_381_fp
#define SF bls12_381_fr
#define SF_N BLS12_381_FR_LIMBS
#define SF_BITS BLS12_381_FR_BITS
#include "curve_tmpl.h"
#undef G
#undef CF

#define G bls12_381_g2
#define CF bls12_381_e2
#include "curve_tmpl.h"
#undef G
#undef CF
#undef SF
#undef SF_N
#undef SF_BITS
#undef G_CS

#define G_CS CS_BW6
#define G bw6_761_g1
#define CF bw6_761_fp
#define SF bw6_761_fr
#define SF_N BW6_761_FR_LIMBS
#define SF_BITS BW6_761_FR_BITS
#include "curve_tmpl.h"
#undef G

#define G bw6_761_g2
#include "curve_tmpl.h"
#undef G
#undef CF
#undef SF
#undef SF_N
#undef SF_BITS
#undef G_CS

/* ------------------------------------------------------------------ fr/fft (fft_tmpl.h), one instance per scalar field */
#define SF bn254_fr
#define SF_ROOT bn254_fr_root_of_unity
#define SF_MAXORD BN254_FR_MAX_ORDER
#define SF_MULTGEN bn254_fr_mult_gen
#include "fft_tmpl.h"
#undef SF
#undef SF_ROOT
#undef SF_MAXORD
#undef SF_MULTGEN
#define SF bls12_381_fr
#define SF_ROOT bls12_381_fr_root_of_unity
#define SF_MAXORD BLS12_381_FR_MAX_ORDER
#define SF_MULTGEN bls12_381_fr_mult_gen
#include "fft_tmpl.h"
#undef SF
#undef SF_ROOT
#undef SF_MAXORD
#undef SF_MULTGEN
#define SF bw6_761_fr
#define SF_ROOT bw6_761_fr_root_of_unity
#define SF_MAXORD BW6_761_FR_MAX_ORDER
#define SF_MULTGEN bw6_761_fr_mult_gen
#include "fft_tmpl.h"
#undef SF
#undef SF_ROOT
#undef SF_MAXORD
#undef SF_MULTGEN

/* ------------------------------------------------------------------ exported C API (ctypes) */
#define EXPORT __attribute__((visibility("default")))

#include <time.h>
#define bn254_fp_RSQ_ARR bn254_fp_rsquare
#define bn254_fr_RSQ_ARR bn254_fr_rsquare
#define bls12_381_fp_RSQ_ARR bls12_381_fp_rsquare
#define bls12_381_fr_RSQ_ARR bls12_381_fr_rsquare
#define bw6_761_fp_RSQ_ARR bw6_761_fp_rsquare
#define bw6_761_fr_RSQ_ARR bw6_761_fr_rsquare
/* 1 (default): windows use the batch-affine buckets where the reference would; 0: extended-Jacobian buckets everywhere */
EXPORT void oracle_set_batch_affine(int on) { oracle_batch_affine_on = on; }

#define FIELD_API(F)                                                                                                        \
    EXPORT void oracle_##F##_mul(const uint64_t *a, const uint64_t *b, uint64_t *z) { F##_mul((F##_t *)z, (const F##_t *)a, (const F##_t *)b); } \
    EXPORT void oracle_##F##_add(const uint64_t *a, const uint64_t *b, uint64_t *z) { F##_add((F##_t *)z, (const F##_t *)a, (const F##_t *)b); } \
    EXPORT void oracle_##F##_sub(const uint64_t *a, const uint64_t *b, uint64_t *z) { F##_sub((F##_t *)z, (const F##_t *)a, (const F##_t *)b); } \
    EXPORT void oracle_##F##_neg(const uint64_t *a, uint64_t *z) { F##_neg((F##_t *)z, (const F##_t *)a); }                  \
    EXPORT void oracle_##F##_dbl(const uint64_t *a, uint64_t *z) { F##_dbl((F##_t *)z, (const F##_t *)a); }                  \
    EXPORT void oracle_##F##_sqr(const uint64_t *a, uint64_t *z) { F##_sqr((F##_t *)z, (const F##_t *)a); }                  \
    EXPORT void oracle_##F##_inv(const uint64_t *a, uint64_t *z) { F##_inv((F##_t *)z, (const F##_t *)a); }

#define PRIME_FIELD_API(F)                                                                                                  \
    FIELD_API(F)                                                                                                            \
    EXPORT void oracle_##F##_from_mont(const uint64_t *a, uint64_t *z) { F##_t t = *(const F##_t *)a; F##_from_mont(&t); *(F##_t *)z = t; } \
    EXPORT void oracle_##F##_to_mont(const uint64_t *a, uint64_t *z) { F##_to_mont((F##_t *)z, (const F##_t *)a); }     \
    /* nanoseconds per Montgomery product on this host (dependent chain of 2*iters products, one core) */             \
    EXPORT double oracle_##F##_mul_ns(unsigned iters) {                                                                 \
        F##_t a, b; F##_set_one(&a); memcpy(b.l, F##_RSQ_ARR, sizeof b.l);                                              \
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);                                                    \
        for (unsigned i = 0; i < iters; ++i) { F##_mul(&a, &a, &b); F##_mul(&b, &b, &a); }                              \
        clock_gettime(CLOCK_MONOTONIC, &t1);                                                                            \
        volatile uint64_t sink = a.l[0] ^ b.l[0]; (void)sink;                                                           \
        return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / (2.0 * iters); }           \
    EXPORT void oracle_##F##_mul_generic(const uint64_t *a, const uint64_t *b, uint64_t *z) { F##_mul_generic((F##_t *)z, (const F##_t *)a, (const F##_t *)b); } \
    /* z = sum_i a[i]*b[i] (Montgomery products, so z is the Montgomery form of sum a_i b_i): closed-form check of an   \
     * MSM over bases [a_i]G with scalars b_i, whose result must be [sum a_i b_i]G */                                        \
    EXPORT void oracle_##F##_dot(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *z) {                             \
        F##_t acc, t; F##_set_zero(&acc);                                                                                   \
        for (size_t i = 0; i < n; ++i) { F##_mul(&t, (const F##_t *)a + i, (const F##_t *)b + i); F##_add(&acc, &acc, &t); } \
        *(F##_t *)z = acc; }

#define FFT_API(F)                                                                                                          \
    EXPORT int oracle_##F##_fft_generator(uint64_t m, uint64_t *gen) { return F##_fft_generator(m, (F##_t *)gen); }           \
    EXPORT int oracle_##F##_fft(uint64_t *a, size_t n, int inverse, int decimation, int coset) {                             \
        return F##_fft_transform((F##_t *)a, n, inverse, decimation, coset); }                                               \
    EXPORT void oracle_##F##_fft_bit_reverse(uint64_t *a, size_t n) { F##_fft_bit_reverse((F##_t *)a, n); }
FFT_API(bn254_fr)
FFT_API(bls12_381_fr)
FFT_API(bw6_761_fr)

PRIME_FIELD_API(bn254_fp)
PRIME_FIELD_API(bn254_fr)
PRIME_FIELD_API(bls12_381_fp)
PRIME_FIELD_API(bls12_381_fr)
PRIME_FIELD_API(bw6_761_fp)
PRIME_FIELD_API(bw6_761_fr)
FIELD_API(bn254_e2)
FIELD_API(bls12_381_e2)

#define GROUP_API(GG, SFLD)                                                                                                 \
    /* p (XYZZ, in/out) += (negate ? -a : a) */                                                                             \
    EXPORT void oracle_##GG##_xyzz_add_mixed(uint64_t *p, const uint64_t *a, int negate) {                                  \
        GG##_xyzz_add_mixed((GG##_xyzz_t *)p, (const GG##_aff_t *)a, negate); }                                             \
    EXPORT void oracle_##GG##_xyzz_add(uint64_t *p, const uint64_t *q) { GG##_xyzz_add((GG##_xyzz_t *)p, (const GG##_xyzz_t *)q); } \
    EXPORT void oracle_##GG##_xyzz_double(uint64_t *p, const uint64_t *q) { GG##_xyzz_double((GG##_xyzz_t *)p, (const GG##_xyzz_t *)q); } \
    EXPORT void oracle_##GG##_xyzz_set_infinity(uint64_t *p) { GG##_xyzz_set_infinity((GG##_xyzz_t *)p); }                  \
    EXPORT void oracle_##GG##_xyzz_to_jac(const uint64_t *p, uint64_t *j) { GG##_jac_from_xyzz((GG##_jac_t *)j, (const GG##_xyzz_t *)p); } \
    EXPORT void oracle_##GG##_jac_add_assign(uint64_t *p, const uint64_t *q) { GG##_jac_add_assign((GG##_jac_t *)p, (const GG##_jac_t *)q); } \
    EXPORT void oracle_##GG##_jac_to_affine(const uint64_t *j, uint64_t *a) { GG##_aff_from_jac((GG##_aff_t *)a, (const GG##_jac_t *)j); } \
    /* [k]a, k = plain little-endian limbs; result Jacobian */                                                              \
    EXPORT void oracle_##GG##_scalar_mul(const uint64_t *a, const uint64_t *k, int klimbs, uint64_t *out_jac) {             \
        GG##_xyzz_t r; GG##_scalar_mul(&r, (const GG##_aff_t *)a, k, klimbs); GG##_jac_from_xyzz((GG##_jac_t *)out_jac, &r); } \
    /* digits[chunk*n+i], uint16, caller-zeroed, nb_chunks(c)*n entries */                                                  \
    EXPORT unsigned oracle_##GG##_nb_chunks(unsigned c) { return GG##_nb_chunks(c); }                                       \
    EXPORT void oracle_##GG##_partition_scalars(const uint64_t *scalars, size_t n, unsigned c, uint16_t *digits) {          \
        GG##_partition_range((const SFLD##_t *)scalars, n, c, digits, 0, n); }                                              \
    /* one window: points, that window's digits -> XYZZ total */                                                            \
    EXPORT void oracle_##GG##_process_chunk(unsigned c, const uint64_t *points, const uint16_t *digits, size_t n, uint64_t *out_xyzz) { \
        GG##_xyzz_t *b = (GG##_xyzz_t *)malloc(sizeof(GG##_xyzz_t) << (c - 1));                                             \
        GG##_process_chunk(c, (const GG##_aff_t *)points, digits, n, b, (GG##_xyzz_t *)out_xyzz); free(b); }                \
    /* _innerMsm with a fixed window size c */                                                                              \
    EXPORT void oracle_##GG##_msm_c(const uint64_t *points, const uint64_t *scalars, size_t n, unsigned c, int nthreads, uint64_t *out_jac) { \
        GG##_inner_msm((GG##_jac_t *)out_jac, c, (const GG##_aff_t *)points, (const SFLD##_t *)scalars, n, nthreads); }     \
    /* (*Jac).MultiExp: bestC + split recursion; returns 0 / 1 (len mismatch) / 2 (NbTasks > 1024) */                       \
    EXPORT int oracle_##GG##_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,   \
                                      int nb_tasks, int num_cpu, int nthreads, uint64_t *out_jac) {                         \
        return GG##_multiexp((GG##_jac_t *)out_jac, (const GG##_aff_t *)points, n_points, (const SFLD##_t *)scalars, n_scalars, nb_tasks, num_cpu, nthreads); } \
    EXPORT unsigned oracle_##GG##_best_c(size_t n) { return GG##_best_c(n); }                                               \
    /* out[i] = [k0 + i*k1] base, i < n; k0,k1 plain limbs */                                                               \
    EXPORT void oracle_##GG##_gen_points(const uint64_t *base, const uint64_t *k0, const uint64_t *k1, int klimbs, size_t n, int nthreads, uint64_t *out) { \
        if (nthreads < 1) nthreads = 1;                                                                                     \
        GG##_xyzz_t sx; GG##_aff_t step;                                                                                    \
        GG##_scalar_mul(&sx, (const GG##_aff_t *)base, k1, klimbs); GG##_aff_from_xyzz(&step, &sx);                         \
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);                                          \
        GG##_gen_t *jobs = (GG##_gen_t *)malloc(sizeof(GG##_gen_t) * (size_t)nthreads);                                     \
        size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;                                                         \
        int used = 0;                                                                                                       \
        for (int t = 0; t < nthreads; ++t) {                                                                                \
            size_t s = (size_t)t * per, e = s + per < n ? s + per : n;                                                      \
            if (s >= e) break;                                                                                              \
            jobs[t] = (GG##_gen_t){(const GG##_aff_t *)base, &step, k0, k1, klimbs, s, e, (GG##_aff_t *)out};               \
            pthread_create(&th[t], NULL, GG##_gen_worker, &jobs[t]); ++used; }                                              \
        for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);                                                           \
        free(th); free(jobs); }

GROUP_API(bn254_g1, bn254_fr)
GROUP_API(bn254_g2, bn254_fr)
GROUP_API(bls12_381_g1, bls12_381_fr)
GROUP_API(bls12_381_g2, bls12_381_fr)
GROUP_API(bw6_761_g1, bw6_761_fr)
GROUP_API(bw6_761_g2, bw6_761_fr)
