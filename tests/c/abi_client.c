/* A plain-C client of include/gmsm.h, standing in for the cgo stub of INTEGRATION.md: it binds the drop-in symbols the
 * way the Go side would (same argument order, raw pointers to the slices' backing arrays) and talks to nothing else.
 *
 *   abi_client <group 0..5> <in.bin> <out.bin>
 *   in.bin  : u64 n | u64 aff_limbs | u64 fr_limbs | n*aff_limbs u64 points | n*fr_limbs u64 scalars   (little endian,
 *             the layout of utils/unsafe.WriteSlice payloads, utils/unsafe/dump_slice.go:16-32)
 *   out.bin : 4 blocks of aff_limbs u64: MultiExp via the per-curve drop-in symbol (Jacobian -> affine with
 *             gmsm_jac_to_affine), via gmsm_multiexp_affine, via registered bases, and via bases that went through the
 *             Encoder's default wire format (gmsm_points_compress -> gmsm_bases_register_compressed with the Decoder's
 *             default checks: what kzg's ReadFromResident does, integration/go); then u64 rc of the length-mismatch call
 *             and u64 rc of the NbTasks > 1024 call.
 * Exit code 0 when every call returned what the contract says. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gmsm.h"

typedef int (*dropin_fn)(const uint64_t *, size_t, const uint64_t *, size_t, int, uint64_t *);

int main(int argc, char **argv) {
    if (argc != 4) return 64;
    const int group = atoi(argv[1]);
    static const dropin_fn dropin[6] = {gmsm_bn254_g1_multiexp,     gmsm_bn254_g2_multiexp,   gmsm_bls12_381_g1_multiexp,
                                        gmsm_bls12_381_g2_multiexp, gmsm_bw6_761_g1_multiexp, gmsm_bw6_761_g2_multiexp};
    if (group < 0 || group > 5) return 64;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 65;
    uint64_t hdr[3];
    if (fread(hdr, 8, 3, f) != 3) return 65;
    const size_t n = (size_t)hdr[0], al = (size_t)hdr[1], fl = (size_t)hdr[2];
    if (al != gmsm_affine_limbs(group) || fl != gmsm_scalar_limbs(group)) return 66;
    uint64_t *pts = malloc((n * al + 1) * 8), *sc = malloc((n * fl + 1) * 8);
    if (fread(pts, 8, n * al, f) != n * al || fread(sc, 8, n * fl, f) != n * fl) return 65;
    fclose(f);

    uint64_t jac[3 * 24], aff[4][2 * 24];
    memset(aff, 0, sizeof aff);
    int rc = dropin[group](pts, n, sc, n, 0, jac);
    if (rc != GMSM_OK) { fprintf(stderr, "drop-in: %d %s\n", rc, gmsm_last_error()); return 1; }
    if (gmsm_jac_to_affine(group, jac, aff[0]) != GMSM_OK) return 2;
    if (gmsm_multiexp_affine(group, pts, n, sc, n, 8, aff[1]) != GMSM_OK) return 3;
    uint64_t h = 0;
    if (gmsm_bases_register(group, pts, NULL, n, &h) != GMSM_OK) { fprintf(stderr, "%s\n", gmsm_last_error()); return 4; }
    if (gmsm_multiexp_bases(h, sc, n, 0, jac) != GMSM_OK) return 5;
    if (gmsm_jac_to_affine(group, jac, aff[2]) != GMSM_OK) return 6;
    if (gmsm_bases_release(h) != GMSM_OK) return 7;
    {   /* the same bases as compressed bytes: X and the flag of Y's half, al * 4 bytes a point */
        uint8_t *comp = malloc(n * al * 4 + 1);
        int64_t bad = -1;
        if (gmsm_points_compress(group, pts, NULL, n, comp) != GMSM_OK) { fprintf(stderr, "compress: %s\n", gmsm_last_error()); return 9; }
        if (gmsm_bases_register_compressed(group, comp, n, 2, &h, &bad) != GMSM_OK) { fprintf(stderr, "register_compressed: point %lld: %s\n", (long long)bad, gmsm_last_error()); return 10; }
        if (gmsm_multiexp_bases(h, sc, n, 0, jac) != GMSM_OK) return 11;
        if (gmsm_jac_to_affine(group, jac, aff[3]) != GMSM_OK) return 12;
        if (gmsm_bases_release(h) != GMSM_OK) return 13;
        free(comp);
    }
    const uint64_t rc_len = (uint64_t)dropin[group](pts, n, sc, n ? n - 1 : 1, 0, jac);
    const uint64_t rc_cfg = (uint64_t)dropin[group](pts, n, sc, n, 1025, jac);

    f = fopen(argv[3], "wb");
    if (!f) return 65;
    for (int i = 0; i < 4; ++i) fwrite(aff[i], 8, al, f);
    fwrite(&rc_len, 8, 1, f);
    fwrite(&rc_cfg, 8, 1, f);
    fclose(f);
    return (rc_len == GMSM_ERR_LEN && rc_cfg == GMSM_ERR_CONFIG) ? 0 : 8;
}
