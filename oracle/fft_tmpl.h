/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product path.
 *
 * fr/fft template: CPU restatement of gnark-crypto's FFT over one scalar field. Define before including:
 *   SF          scalar-field prefix (an fp_tmpl.h instance, e.g. bn254_fr)
 *   SF_ROOT     Montgomery limbs of the primitive 2^SF_MAXORD-th root of unity (fr/generator.go:23)
 *   SF_MAXORD   fr/generator.go:24
 *   SF_MULTGEN  Montgomery limbs of the generator of Fr^* used as coset shift (fr/fft/domain.go:56-62)
 *
 * Follows ecc/bn254/fr/fft:
 *   fr.Generator            fr/generator.go:18-36
 *   NewDomain               fft/domain.go:66-99   (Generator, GeneratorInv, CardinalityInv, FrMultiplicativeGen(Inv))
 *   (*Domain).FFT           fft/fft.go:31-113     (coset scaling first; DIT reads the coset table bit-reversed)
 *   (*Domain).FFTInverse    fft/fft.go:115-196    (scaling by CardinalityInv and the inverse coset table afterwards)
 *   difFFT / ditFFT         fft/fft.go:198-262 / :285-360, the recursion on halves with w squared per level
 *                           (the "without twiddles" forms: the twiddle of butterfly i is w^i, computed by running product
 *                           like innerDIFWithoutTwiddles :275-284; precomputed tables hold the same values)
 *   BitReverse              fft/bitreverse.go:33-45 (naive form; the cobra variant computes the same permutation)
 */
#define FFF(name) ORACLE_CAT(SF, ORACLE_CAT(_fft_, name))
#define FSF(name) ORACLE_CAT(SF, ORACLE_CAT(_, name))
#define FST ORACLE_CAT(SF, _t)

static void FFF(pow_u64)(FST *z, const FST *x, uint64_t e) {
    FST acc, base = *x;
    FSF(set_one)(&acc);
    while (e) {
        if (e & 1) FSF(mul)(&acc, &acc, &base);
        FSF(sqr)(&base, &base);
        e >>= 1;
    }
    *z = acc;
}

/* fr.Generator(m): generator of the subgroup of order NextPowerOfTwo(m); returns 0 on success */
static int FFF(generator)(uint64_t m, FST *gen) {
    unsigned logx = 0;
    while (logx < 63 && ((uint64_t)1 << logx) < m) ++logx;
    if (logx > SF_MAXORD) return 1;
    FST root;
    memcpy(root.l, SF_ROOT, sizeof root.l);
    FFF(pow_u64)(gen, &root, (uint64_t)1 << (SF_MAXORD - logx));
    return 0;
}

static void FFF(dif)(FST *a, size_t n, FST w) {  /* difFFT, fft.go:198 */
    if (n == 1) return;
    const size_t m = n >> 1;
    FST at = w, t;
    /* innerDIFWithoutTwiddles, fft.go:275: butterfly 0 needs no twiddle */
    FSF(add)(&t, &a[0], &a[m]);
    FSF(sub)(&a[m], &a[0], &a[m]);
    a[0] = t;
    for (size_t i = 1; i < m; ++i) {
        FSF(add)(&t, &a[i], &a[i + m]);       /* fr.Butterfly: (a, b) <- (a + b, a - b) */
        FSF(sub)(&a[i + m], &a[i], &a[i + m]);
        a[i] = t;
        FSF(mul)(&a[i + m], &a[i + m], &at);
        FSF(mul)(&at, &at, &w);
    }
    if (m == 1) return;
    FSF(sqr)(&w, &w);
    FFF(dif)(a, m, w);
    FFF(dif)(a + m, m, w);
}

static void FFF(dit)(FST *a, size_t n, FST w) {  /* ditFFT, fft.go:285 */
    if (n == 1) return;
    const size_t m = n >> 1;
    FST next;
    FSF(sqr)(&next, &w);
    FFF(dit)(a, m, next);
    FFF(dit)(a + m, m, next);
    /* innerDITWithoutTwiddles: a[i+m] *= w^i, then butterfly */
    FST at = w, t;
    FSF(add)(&t, &a[0], &a[m]);
    FSF(sub)(&a[m], &a[0], &a[m]);
    a[0] = t;
    for (size_t i = 1; i < m; ++i) {
        FSF(mul)(&a[i + m], &a[i + m], &at);
        FSF(add)(&t, &a[i], &a[i + m]);
        FSF(sub)(&a[i + m], &a[i], &a[i + m]);
        a[i] = t;
        FSF(mul)(&at, &at, &w);
    }
}

static size_t FFF(rev)(size_t i, unsigned logn) {
    size_t r = 0;
    for (unsigned b = 0; b < logn; ++b) r |= ((i >> b) & 1) << (logn - 1 - b);
    return r;
}

static void FFF(bit_reverse)(FST *a, size_t n) {  /* bitreverse.go:33-45 */
    unsigned logn = 0;
    while (((size_t)1 << logn) < n) ++logn;
    for (size_t i = 0; i < n; ++i) {
        const size_t r = FFF(rev)(i, logn);
        if (r > i) { FST t = a[i]; a[i] = a[r]; a[r] = t; }
    }
}

/* (*Domain).FFT (inverse == 0) / FFTInverse on a vector of n = cardinality elements; decimation 0 = DIT, 1 = DIF.
 * Returns 0, or 1 when n is not a power of two within the field's 2-adicity. */
static int FFF(transform)(FST *a, size_t n, int inverse, int decimation, int coset) {
    unsigned logn = 0;
    while (((size_t)1 << logn) < n) ++logn;
    if (((size_t)1 << logn) != n) return 1;
    FST gen, gen_inv, shift, shift_inv, card, card_inv;
    if (FFF(generator)((uint64_t)n, &gen)) return 1;
    FSF(inv)(&gen_inv, &gen);
    memcpy(shift.l, SF_MULTGEN, sizeof shift.l);
    FSF(inv)(&shift_inv, &shift);
    FSF(set_one)(&card);
    for (unsigned i = 0; i < logn; ++i) FSF(dbl)(&card, &card);  /* SetUint64(n) */
    FSF(inv)(&card_inv, &card);
    if (!inverse) {
        if (coset) {  /* fft.go:43-82 */
            FST *table = (FST *)malloc(sizeof(FST) * n);
            FSF(set_one)(&table[0]);
            for (size_t i = 1; i < n; ++i) FSF(mul)(&table[i], &table[i - 1], &shift);  /* BuildExpTable */
            for (size_t i = 0; i < n; ++i) {
                const size_t t = decimation == 0 ? FFF(rev)(i, logn) : i;  /* DIT: input is bit-reversed */
                FSF(mul)(&a[i], &a[i], &table[t]);
            }
            free(table);
        }
        if (decimation == 1) FFF(dif)(a, n, gen); else FFF(dit)(a, n, gen);
        return 0;
    }
    if (decimation == 1) FFF(dif)(a, n, gen_inv); else FFF(dit)(a, n, gen_inv);
    if (!coset) {  /* fft.go:144-151 */
        for (size_t i = 0; i < n; ++i) FSF(mul)(&a[i], &a[i], &card_inv);
        return 0;
    }
    FST *table = (FST *)malloc(sizeof(FST) * n);
    FSF(set_one)(&table[0]);
    for (size_t i = 1; i < n; ++i) FSF(mul)(&table[i], &table[i - 1], &shift_inv);
    for (size_t i = 0; i < n; ++i) {  /* fft.go:153-195: DIT output natural, DIF output bit-reversed */
        const size_t t = decimation == 1 ? FFF(rev)(i, logn) : i;
        FSF(mul)(&a[i], &a[i], &table[t]);
        FSF(mul)(&a[i], &a[i], &card_inv);
    }
    free(table);
    return 0;
}

#undef FFF
#undef FSF
#undef FST
