/* Race / memory stress of libgmsm's host shim, meant to run against the sanitizer builds (`make -C gnark-crypto_amd/csrc
 * tsan` / `asan`; the reference's CI runs `go test -race ./ecc/bn254/...`, .github/workflows/pr.yml:63).  Plain C +
 * pthreads, links only the library: NTHREADS threads hammer every kind of entry at once - the blocking drop-in call, the
 * registered-bases call, submit/collect tickets, handle registration and release under use, gmsm_bases_precompute while
 * other threads run MultiExp over the same handle, small calls (the fused kernel's per-workspace counters), first-use coset
 * FFTs on a shared domain, gmsm_trim and the option
 * switches - and every result must equal the one computed single-threaded before the race started.
 *
 *   race_client [iterations per thread, default 6] [n, default 20000]
 * Exit code 0 = every result equal and every call returned GMSM_OK (the sanitizer adds its own verdict on stderr). */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gmsm.h"

#define NTHREADS 6
#define G GMSM_BN254_G1
#define AL 8
#define FL 4

static size_t N = 20000, ITER = 6;
static uint64_t *pts, *sc, *fft_in, *fft_ref;
static uint8_t *comp; /* the bases in the Encoder's default (compressed) wire format */
static uint64_t ref_aff[AL], ref_small_aff[AL], shared_handle, fft_domain;
static const size_t N_SMALL = 700; /* a call the fused small-n kernel takes (three slices per window) */
static const size_t FFT_N = 1 << 12;
static int failures = 0;
static pthread_mutex_t fail_mu = PTHREAD_MUTEX_INITIALIZER;

static void bad(const char *what, int rc) {
    pthread_mutex_lock(&fail_mu);
    ++failures;
    fprintf(stderr, "race_client: %s (rc %d): %s\n", what, rc, gmsm_last_error());
    pthread_mutex_unlock(&fail_mu);
}

static void check_jac(const char *what, int rc, const uint64_t *jac) {
    uint64_t aff[AL];
    if (rc != GMSM_OK) { bad(what, rc); return; }
    gmsm_jac_to_affine(G, jac, aff);
    if (memcmp(aff, ref_aff, sizeof aff) != 0) bad(what, -1);
}

static void check_small(const char *what, int rc, const uint64_t *jac) {
    uint64_t aff[AL];
    if (rc != GMSM_OK) { bad(what, rc); return; }
    gmsm_jac_to_affine(G, jac, aff);
    if (memcmp(aff, ref_small_aff, sizeof aff) != 0) bad(what, -1);
}

static void *worker(void *arg) {
    const int id = (int)(intptr_t)arg;
    uint64_t jac[3 * 4];
    for (size_t it = 0; it < ITER; ++it) {
        switch ((id + it) % 6) {
            case 0:
                check_jac("drop-in", gmsm_bn254_g1_multiexp(pts, N, sc, N, 0, jac), jac);
                check_small("drop-in, small", gmsm_bn254_g1_multiexp(pts, N_SMALL, sc, N_SMALL, 1, jac), jac);
                break;
            case 1:
                check_jac("bases", gmsm_multiexp_bases(shared_handle, sc, N, 0, jac), jac);
                check_small("bases, small prefix", gmsm_multiexp_bases(shared_handle, sc, N_SMALL, 1, jac), jac);
                break;
            case 2: {  /* a handle of its own, released while its own call has just finished and others keep going */
                uint64_t h = 0;
                int64_t bad_index = -1;
                /* every other time from the compressed bytes: decode (a square root per point) + subgroup step + register */
                int rc = (it & 1) ? gmsm_bases_register_compressed(G, comp, N, 2, &h, &bad_index) : gmsm_bases_register(G, pts, NULL, N, &h);
                if (rc) { bad("register", rc); break; }
                check_jac("own bases", gmsm_multiexp_bases(h, sc, N, 0, jac), jac);
                if ((rc = gmsm_bases_release(h))) bad("release", rc);
                break;
            }
            case 3: {  /* window tables appear on the shared handle while other threads use it */
                int rc = gmsm_bases_precompute(shared_handle, 0);
                if (rc) bad("precompute", rc);
                check_jac("bases after precompute", gmsm_multiexp_bases(shared_handle, sc, N, 0, jac), jac);
                break;
            }
            case 4: {  /* first-use coset tables of the shared domain, forward and inverse */
                uint64_t *a = malloc(FFT_N * FL * 8);
                memcpy(a, fft_in, FFT_N * FL * 8);
                int rc = gmsm_fft(fft_domain, a, NULL, FFT_N, 0, 1, 1, NULL);
                if (rc) bad("coset fft", rc);
                else if (memcmp(a, fft_ref, FFT_N * FL * 8) != 0) bad("coset fft result", -1);
                if (!rc && (rc = gmsm_fft(fft_domain, a, NULL, FFT_N, 1, 0, 1, NULL))) bad("coset inverse", rc);
                else if (memcmp(a, fft_in, FFT_N * FL * 8) != 0) bad("coset round trip", -1);
                free(a);
                break;
            }
            default: {  /* switches and the trimmer under load */
                size_t freed = 0;
                (void)gmsm_get_option(GMSM_OPT_WINDOW_BITS);
                int rc = gmsm_trim(1 << 20, &freed);
                if (rc) bad("trim", rc);
                check_jac("affine entry after trim", gmsm_multiexp(G, pts, N, sc, N, 4, jac), jac);
                break;
            }
        }
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc > 1) ITER = (size_t)atol(argv[1]);
    if (argc > 2) N = (size_t)atol(argv[2]);
    if (gmsm_device_count() < 1) { fprintf(stderr, "race_client: no device\n"); return 77; }
    /* BN254 G1 generator (1, 2) in Montgomery form is produced by the library itself: [k0 + i k1] * (1, 2) */
    static const uint64_t one_mont[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
    uint64_t gen[AL];
    memcpy(gen, one_mont, 32);
    uint64_t two[4];
    int rc = gmsm_debug_field_op(G, 0, 1, one_mont, one_mont, 1, two); /* 1 + 1 */
    if (rc) { fprintf(stderr, "field op: %s\n", gmsm_last_error()); return 1; }
    memcpy(gen + 4, two, 32);
    pts = malloc(N * AL * 8);
    sc = malloc(N * FL * 8);
    const uint64_t k0[4] = {12345, 0, 0, 0}, k1[4] = {0x9e3779b97f4a7c15ull, 77, 0, 0};
    if ((rc = gmsm_generate_points(G, gen, k0, k1, 4, N, 8, pts))) return 2;
    comp = malloc(N * AL * 4);
    if ((rc = gmsm_points_compress(G, pts, NULL, N, comp))) { fprintf(stderr, "compress: %s\n", gmsm_last_error()); return 2; }
    uint64_t x = 0x243f6a8885a308d3ull;
    for (size_t i = 0; i < N * FL; ++i) {  /* xorshift scalars below 2^252 (canonical limbs) */
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        sc[i] = (i % FL == FL - 1) ? (x >> 12) : x;
    }
    uint64_t jac[12];
    if ((rc = gmsm_bn254_g1_multiexp(pts, N, sc, N, 0, jac))) { fprintf(stderr, "reference call: %s\n", gmsm_last_error()); return 3; }
    gmsm_jac_to_affine(G, jac, ref_aff);
    if (N < N_SMALL) { fprintf(stderr, "race_client: n must be at least %zu\n", N_SMALL); return 3; }
    if ((rc = gmsm_bn254_g1_multiexp(pts, N_SMALL, sc, N_SMALL, 1, jac))) { fprintf(stderr, "small reference call: %s\n", gmsm_last_error()); return 3; }
    gmsm_jac_to_affine(G, jac, ref_small_aff);
    if ((rc = gmsm_bases_register(G, pts, NULL, N, &shared_handle))) return 4;
    fft_in = malloc(FFT_N * FL * 8);
    fft_ref = malloc(FFT_N * FL * 8);
    for (size_t i = 0; i < FFT_N * FL; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        fft_in[i] = (i % FL == FL - 1) ? (x >> 12) : x;
    }
    uint64_t d0 = 0;
    if ((rc = gmsm_fft_domain_new(G, FFT_N, &d0))) return 5;
    memcpy(fft_ref, fft_in, FFT_N * FL * 8);
    if ((rc = gmsm_fft(d0, fft_ref, NULL, FFT_N, 0, 1, 1, NULL))) return 6; /* reference result from its own domain */
    gmsm_fft_domain_release(d0);
    if ((rc = gmsm_fft_domain_new(G, FFT_N, &fft_domain))) return 7;      /* the shared one: coset tables not built yet */

    pthread_t th[NTHREADS];
    for (int t = 0; t < NTHREADS; ++t) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
    /* the main thread keeps two tickets in flight meanwhile (needs device scalars: skipped here - the ticket path takes
     * device pointers, which a plain-C client without the HIP headers cannot make; tests/test_gpu_parity.py covers it) */
    for (int t = 0; t < NTHREADS; ++t) pthread_join(th[t], NULL);
    gmsm_bases_release(shared_handle);
    gmsm_fft_domain_release(fft_domain);
    if ((rc = gmsm_shutdown())) { bad("shutdown", rc); }
    /* the library comes back after a shutdown */
    check_jac("after shutdown", gmsm_bn254_g1_multiexp(pts, N, sc, N, 0, jac), jac);
    gmsm_shutdown();
    printf("race_client: %d threads x %zu iterations, n = %zu: %d failure(s)\n", NTHREADS, ITER, N, failures);
    fflush(stdout); /* the verdict must survive whatever the sanitizer and HIP runtimes do while the process is torn down */
    return failures ? 1 : 0;
}
