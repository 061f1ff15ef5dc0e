/* gmsm.h -- C ABI of the MI355X multi-scalar-multiplication engine (libgmsm.so).
 *
 * Drop-in boundary for gnark-crypto's  (*G1Jac).MultiExp / (*G2Jac).MultiExp
 *   ecc/bn254/multiexp.go:32 (G1) and :357 (G2); ecc/bls12-381/multiexp.go:32,:355; ecc/bw6-761/multiexp.go:32,:306
 * A build-tagged Go file per curve package replaces the body of those methods (after the two argument checks,
 * multiexp.go:61-71) with one cgo call to the matching symbol below; (*G1Affine).MultiExp, Fold, kzg.Commit,
 * pedersen etc. route through it unchanged.  INTEGRATION.md shows the cgo stub.
 *
 * Memory layout (identical to the Go slices, so `unsafe.Pointer(&points[0])` is passed without copying):
 *   points   n x {X,Y}           each coordinate = fp.Element  = FP_LIMBS little-endian uint64, Montgomery form;
 *                                G2 over Fp2: X.A0, X.A1, Y.A0, Y.A1.  The affine point (0,0) is infinity.
 *   scalars  n x fr.Element      FR_LIMBS little-endian uint64, Montgomery form.
 *   out_jac  {X,Y,Z}             Jacobian, Montgomery form; Z = 0 (X = Y = 1) for infinity.  Any representative of
 *                                the group element (as in the reference, callers compare with Equal or FromJacobian).
 *   curve      FP_LIMBS FR_LIMBS   G1Affine  G2Affine  (bytes)
 *   bn254         4        4          64       128
 *   bls12_381     6        4          96       192
 *   bw6_761      12        6         192       192     (G2 is over Fp)
 *
 * Return codes:  GMSM_OK, GMSM_ERR_LEN  ("len(points) != len(scalars)", multiexp.go:63),
 *                GMSM_ERR_CONFIG ("invalid config: config.NbTasks > 1024", multiexp.go:70),
 *                >= GMSM_ERR_DEVICE: HIP/runtime failure -- gmsm_last_error() has the text.  There is NO CPU
 *                fallback inside this library: without a usable gfx950 device every compute entry fails loudly.
 *
 * Threading: every entry point is re-entrant and may be called concurrently from any OS thread (goroutines
 * migrate); per device up to three calls run at a time (three workspaces, of which submitted tickets hold at most two, so a
 * blocking entry never waits for somebody else's gmsm_multiexp_collect), further callers wait their turn.
 * Several GPUs: OPT-IN. A process that configures nothing runs every drop-in call on one device (gmsm_set_device, else
 * device 0) and touches no other - the usual deployment is one prover per GPU. After gmsm_set_devices(list) or with
 * GMSM_DEVICES="0,1,2,3" / "all" in the environment (read once) the drop-in entries spread every MultiExp of 2^17 points
 * or more over the listed devices (point slices, one host thread per device, one fold; see gmsm_multiexp_sharded); the
 * footprint is then one context per listed device: six streams, pinned result buffers, scratch that grows with the
 * largest call (gmsm_trim gives it back) and two pooled host threads.
 * Limits: n < 2^31. One pipeline run takes up to 2^27 points; the MultiExp entries split larger inputs into point
 * ranges themselves, gmsm_window_sums_* and gmsm_multiexp_bases_submit refuse them (GMSM_ERR_ARG).
 * Ownership: the caller owns all buffers; host pointers are not retained after return.
 */
#ifndef GMSM_H
#define GMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMSM_OK 0
#define GMSM_ERR_LEN 1
#define GMSM_ERR_CONFIG 2
#define GMSM_ERR_DEVICE 3
#define GMSM_ERR_ARG 4
#define GMSM_ERR_POINT 5 /* a point failed decoding / validation; *bad_index has its position, gmsm_last_error() why */

/* group ids for the generic entry points */
enum gmsm_group {
    GMSM_BN254_G1 = 0,
    GMSM_BN254_G2 = 1,
    GMSM_BLS12_381_G1 = 2,
    GMSM_BLS12_381_G2 = 3,
    GMSM_BW6_761_G1 = 4,
    GMSM_BW6_761_G2 = 5,
    GMSM_NUM_GROUPS = 6
};

/* ---- drop-in entries: host pointers in, Jacobian out (replaces (*GxJac).MultiExp after its argument checks;
 *      the checks are repeated here so that the error behaviour is identical when called directly). ---- */
int gmsm_bn254_g1_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                           int nb_tasks, uint64_t *out_jac); /* ecc/bn254/multiexp.go:32 */
int gmsm_bn254_g2_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                           int nb_tasks, uint64_t *out_jac); /* ecc/bn254/multiexp.go:357 */
int gmsm_bls12_381_g1_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                               int nb_tasks, uint64_t *out_jac); /* ecc/bls12-381/multiexp.go:32 */
int gmsm_bls12_381_g2_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                               int nb_tasks, uint64_t *out_jac); /* ecc/bls12-381/multiexp.go:355 */
int gmsm_bw6_761_g1_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                             int nb_tasks, uint64_t *out_jac); /* ecc/bw6-761/multiexp.go:32 */
int gmsm_bw6_761_g2_multiexp(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                             int nb_tasks, uint64_t *out_jac); /* ecc/bw6-761/multiexp.go:306 */

/* Same, selected by id. */
int gmsm_multiexp(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                  int nb_tasks, uint64_t *out_jac);

/* (*GxAffine).MultiExp (multiexp.go:20-27): MultiExp followed by FromJacobian (g1.go:150-166); out_affine = {X,Y},
 * (0,0) for infinity.  This is the canonical value parity is defined on. */
int gmsm_multiexp_affine(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars,
                         size_t n_scalars, int nb_tasks, uint64_t *out_affine);

/* (*G1Jac).Fold / (*G2Jac).Fold (ecc/bn254/multiexp.go:331-340, G2 :657-665): sum_i points[i] * coeff^i. coeff is one
 * fr.Element (Montgomery limbs); the powers are formed on the host exactly as the reference does, then MultiExp. */
int gmsm_fold(int group, const uint64_t *points, size_t n_points, const uint64_t *combination_coeff, int nb_tasks,
              uint64_t *out_jac);

/* ---- one MultiExp over several GPUs of the node, inside the library.  The reference spreads a
 *      MultiExp over the cores of the machine - one worker per c-bit window collected on a channel
 *      (ecc/bn254/multiexp.go:148-209), and a split of the points in two halves whose results are added (:98-140) -; here
 *      the workers are the devices: one host thread per logical rank runs its piece on its device, the ranks' window totals
 *      (<= 64 extended-Jacobian points each) come back through each device's pinned result buffer and the calling thread
 *      runs the one fold (multiexp.go:302-315).  Same arguments, layouts, return codes and result as gmsm_multiexp.
 *      devices / n_devices: one entry per logical rank, a device may appear more than once (ranks on one device share
 *      its workspaces); NULL / 0 = the configured list (gmsm_set_devices, else GMSM_DEVICES), and every visible device
 *      when nothing is configured - this entry is an explicit request for several devices.  mode: 0 auto (= 1), 1 points - rank r takes points [r n/G, (r+1) n/G) and all windows; every device
 *      copies only its slice over its own PCIe link -, 2 windows - rank r takes windows r, r+G, ... of all points (the
 *      reference's per-window workers; every rank needs all bases).  Ranks whose slice would fall below 2^16 points are
 *      left out.  gmsm_<curve>_g{1,2}_multiexp, gmsm_multiexp, gmsm_multiexp_affine and gmsm_fold call this themselves
 *      when more than one device has been configured (opt-in, see "Several GPUs" above).  Every worker checks that it runs
 *      on the device of its context; a failure names the rank and the device. ---- */
int gmsm_multiexp_sharded(int group, const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                          int nb_tasks, const int *devices, int n_devices, int mode, uint64_t *out_jac);
/* Resident bases on several devices: the n bases are uploaded (all devices at once, each over its own link) and
 * rewritten once on every distinct device of the list - full copies, so that any prefix can be cut evenly over the ranks.
 * The handle is accepted by gmsm_multiexp_bases (host scalars; sharded like gmsm_multiexp_sharded, mode auto),
 * gmsm_multiexp_bases_sharded (explicit mode), gmsm_multiexp_bases_batch (host scalars: the k vectors are independent
 * MultiExp calls, rank r runs the block [r k/G, (r+1) k/G) of them on its device's copy - no sharding of a single call and
 * no exchange; kzg.Commit per polynomial, ecc/bn254/kzg/kzg.go:159-176) and gmsm_bases_release; the device-pointer
 * entries take single-device handles only. */
int gmsm_bases_register_sharded(int group, const uint64_t *points, size_t n, const int *devices, int n_devices,
                                uint64_t *out_handle);
int gmsm_multiexp_bases_sharded(uint64_t handle, const uint64_t *scalars, size_t n_scalars, int nb_tasks, int mode,
                                uint64_t *out_jac);
/* Window tables for registered bases (any handle of gmsm_bases_register* / _sharded): the multiples 2^(c w) P_i of every
 * base for every window w of the c-bit decomposition are computed once and kept in HBM (nwin copies of the bases: BN254 G1,
 * 2^20 bases, c = 19: 14 x 64 MiB - sized for 288 GB).  Every later MultiExp over the handle (all gmsm_multiexp_bases*
 * entries, tickets, batches, point-sharded calls) then drops the digits of ALL windows into ONE set of 2^(c-1) buckets:
 * one bucket reduction instead of nwin, and a wider window (fewer additions) than the plain path can afford.  Same
 * group element, hence the same affine result, as without tables (the reference has no counterpart: its MultiExp takes
 * the bases anew on every call, ecc/bn254/multiexp.go:61; this is what kzg.Commit over a fixed SRS can use,
 * ecc/bn254/kzg/kzg.go:159-176).  c = 0: the library's width for this many bases; 2..20 otherwise.  Calls over a prefix
 * shorter than n/16, call sizes outside the range where the tables were measured to win (about 2^13..2^21 points, by
 * group) and calls with GMSM_OPT_WINDOW_BITS forced to another width use the plain path; GMSM_OPT_TABLES = 0 switches the
 * tables off, 2 uses them for every call size (tests).
 * The same call also builds NARROW tables (width 6) over the first 4096 bases (2048 for BW6-761): a MultiExp of at most that many points over the
 * handle (Pedersen commitments, small KZG commitments over a fixed SRS) then runs the fused small-n kernel with one bucket
 * set per workgroup and needs no host-side fold of window totals at all - BN254 G1, 2^10 points: 0.146 ms against 0.181 ms
 * for the same call without tables.
 * Safe while other threads use the handle: the tables are published (release store of their width) only when complete,
 * calls that started earlier keep the plain path; concurrent precompute calls on one handle are serialised.  Bases outside the prime-order subgroup whose multiples reach
 * the identity are refused (GMSM_ERR_ARG; the handle keeps working without tables).
 * gmsm_bases_table_bits: the width of the handle's tables, 0 = none. */
int gmsm_bases_precompute(uint64_t handle, unsigned c);
unsigned gmsm_bases_table_bits(uint64_t handle);
unsigned long gmsm_debug_table_runs(void); /* pipeline runs that went through window tables so far (tests) */
unsigned long gmsm_debug_small_runs(void); /* calls served by the fused small-n kernel so far (tests) */
/* The devices the drop-in entries shard over (one entry per logical rank); count = 0 restores the default (GMSM_DEVICES
 * if set, else no spreading: one device).  gmsm_get_devices returns the number of configured ranks (1 and the calling
 * thread's device when nothing is configured; -1 when GMSM_DEVICES is malformed) and writes up to max_devices of them
 * (out_devices may be NULL). */
int gmsm_set_devices(const int *devices, int count);
int gmsm_get_devices(int *out_devices, int max_devices);

/* ---- device-resident entries (bases/scalars already in HBM: the SRS-resident fast path).
 *      d_points / d_scalars are device pointers with the layouts above; hip_stream is the hipStream_t the inputs were
 *      produced on (NULL = default stream): the engine runs on a private stream ordered after the work queued there;
 *      the call returns after the result has been copied back to out_jac (host memory).
 *      Concurrency (all blocking entries): any number of threads may call at once; per device up to three calls are in
 *      flight at a time, each on its own workspace and streams - the sort and accumulation of one overlap the
 *      latency-bound reduction, copy-back and host fold of another - and further callers wait their turn. At most two of
 *      the three workspaces are ever held by uncollected tickets (gmsm_multiexp_submit), so a blocking call never waits
 *      for somebody else's gmsm_multiexp_collect. ---- */
int gmsm_multiexp_device(int group, const void *d_points, const void *d_scalars, size_t n, void *hip_stream,
                         uint64_t *out_jac);

/* ---- resident bases (device-resident SRS): the bases of e.g. a KZG proving key
 *      (ecc/bn254/kzg/kzg.go:35-37) are uploaded and rewritten into the engine's internal form once;
 *      every later MultiExp over a prefix of them (kzg.Commit: pk.G1[:len(p)], kzg.go:159-176) sends only scalars.
 *      Give exactly one of `points` (host, Go layout) / `d_points` (device, Go layout). ---- */
int gmsm_bases_register(int group, const uint64_t *points, const void *d_points, size_t n, uint64_t *out_handle);
int gmsm_bases_release(uint64_t handle);
/* scalars on the host (n_scalars <= registered n) */
int gmsm_multiexp_bases(uint64_t handle, const uint64_t *scalars, size_t n_scalars, int nb_tasks, uint64_t *out_jac);
/* scalars already on the device */
int gmsm_multiexp_bases_device(uint64_t handle, const void *d_scalars, size_t n_scalars, void *hip_stream,
                               uint64_t *out_jac);

/* ---- two MultiExp calls in flight (the reference's equivalent is several goroutines calling
 *      MultiExp at once, BenchmarkManyMultiExpG1Reference, ecc/bn254/multiexp_test.go:385-415).
 *      submit launches the whole device pipeline for device-resident scalars over registered bases and returns without
 *      waiting for the device; collect waits, folds the windows and writes the Jacobian result. At most two tickets may be
 *      outstanding per device: a third submit returns GMSM_ERR_ARG at once. A device has three workspaces; when all of
 *      them are leased at that moment (blocking callers in other threads, gmsm_trim) submit waits for the first to be
 *      released, as a blocking entry would - that wait always ends, because tickets never hold more than two of the
 *      three. A caller still waiting when gmsm_shutdown runs gets GMSM_ERR_DEVICE. d_scalars must stay valid until its
 *      ticket is collected.
 *      hip_stream: the stream the scalars were produced on (NULL = the default stream); the pipeline is ordered
 *      after the work already queued there. The sort/accumulate of one call overlaps the latency-bound bucket reduction, copy-back and
 *      host fold of the other. ---- */
int gmsm_multiexp_bases_submit(uint64_t handle, const void *d_scalars, size_t n_scalars, void *hip_stream,
                               uint64_t *out_ticket);
int gmsm_multiexp_collect(uint64_t ticket, uint64_t *out_jac);
/* (a ticket takes at most 2^27 points, and so does every vector of gmsm_multiexp_bases_batch below: submit refuses more
 * with GMSM_ERR_ARG; the blocking single-vector entries split larger inputs themselves)
 * k MultiExp over the same registered bases in one blocking call (one commitment per polynomial with a fixed SRS:
 * kzg.Commit, ecc/bn254/kzg/kzg.go:159-176; BatchOpenSinglePoint :246): scalars = k x n fr.Element, contiguous, either on
 * the host (`scalars`) or on the device (`d_scalars`, produced on hip_stream); out_jac = k Jacobian results. Two of
 * the k calls are in flight at a time; host scalars of vector i+1 are copied while vector i is being accumulated. */
int gmsm_multiexp_bases_batch(uint64_t handle, const uint64_t *scalars, const void *d_scalars, size_t n, size_t k,
                              void *hip_stream, uint64_t *out_jac);

/* ---- fixed-base batch: BatchScalarMultiplicationG1/G2 (ecc/bn254/g1.go:1039-1118; the step that
 *      builds an SRS, kzg.NewSRS) and BatchJacobianToAffineG1 (g1.go:989-1035). out[i] = scalars[i] * base in affine
 *      coordinates (unique, so results are bit-comparable); scalars are Montgomery fr.Element as everywhere else.
 *      On the device every scalar costs nwin mixed additions from a table of all windows' multiples (no doublings), then
 *      one shared inversion per 32 results. The _device variant reads d_scalars (produced on hip_stream) and writes
 *      n Go-layout affine points to d_out_affine; it returns when they are complete. ---- */
int gmsm_batch_scalar_mul(int group, const uint64_t *base_affine, const uint64_t *scalars, size_t n, uint64_t *out_affine);
int gmsm_batch_scalar_mul_device(int group, const uint64_t *base_affine, const void *d_scalars, size_t n, void *hip_stream,
                                 void *d_out_affine);
/* jac = n x {X, Y, Z} (Go G1Jac/G2Jac layout), Z = 0 -> (0, 0) */
int gmsm_batch_jac_to_affine(int group, const uint64_t *jac, size_t n, uint64_t *out_affine);

/* ---- point ingest: the step before the MSM - points arrive as bytes and must become validated
 *      Montgomery limbs in HBM.  `check`: 0 decode only, 1 + on the curve, 2 + in the r-torsion (the Decoder's default,
 *      ecc/bn254/marshal.go:250-275), decided by the reference's own endomorphism identities (IsInSubGroup:
 *      ecc/bls12-381/g1.go:481-492, g2.go:484-491, ecc/bn254/g2.go:483-497, ecc/bw6-761/g1.go:482-496), 3 = the same
 *      predicate decided as the definition says, "on the curve and [r]P = infinity" (2-6 x the group operations; the
 *      cross-check of level 2).  On a bad point the call returns GMSM_ERR_POINT,
 *      *bad_index is the first offender and gmsm_last_error() carries the reference's error text; otherwise *bad_index = -1.
 *
 *      gmsm_points_from_raw: raw = n points in the uncompressed wire format of (*G1Affine).RawBytes() /
 *      Encoder(RawEncoding()) (marshal.go:826, G2 :1078; bls12-381/marshal.go:855): big-endian, regular (non-Montgomery)
 *      form, X | Y (Fp2: A1 | A0), metadata in the top bits of the first byte, n * gmsm_affine_limbs(group) * 8 bytes in
 *      total.  Compressed encodings are refused here (gmsm_points_from_compressed below).  Decoded points go to out_affine (host) and/or d_out_affine (device);
 *      give at least one.
 *      gmsm_points_validate: the same checks over points that are already Go-layout limbs (host or device).
 *      gmsm_bases_register_raw: decode + check + register in one call (the SRS never exists on the host in limb form).
 *      gmsm_bases_register_dump: an SRS dump written by kzg.SRS.WriteDump (ecc/bn254/kzg/marshal.go:65-95) ends in
 *      unsafe.WriteMarker + unsafe.WriteSlice(pk.G1) = u64 0xdeadbeef | u64 length | raw []G1Affine memory
 *      (utils/unsafe/dump_slice.go:16-32, :80).  `offset` is the byte position of the marker (expect_marker != 0) or of
 *      the length word; at most max_points (0 = all) are loaded, like ReadDump's maxPkPoints.  The file is streamed
 *      through pinned buffers straight into HBM and registered; *out_n receives the number of bases.  ReadDump validates
 *      nothing; check > 0 runs the validation kernel over the loaded points. ---- */
int gmsm_points_from_raw(int group, const uint8_t *raw, size_t n, int check, uint64_t *out_affine, void *d_out_affine,
                         int64_t *bad_index);
int gmsm_points_validate(int group, const uint64_t *points, const void *d_points, size_t n, int check, int64_t *bad_index);
int gmsm_bases_register_raw(int group, const uint8_t *raw, size_t n, int check, uint64_t *out_handle, int64_t *bad_index);
/* The Encoder's DEFAULT format - compressed points, what (*G1Affine).Bytes() writes and kzg.SRS.WriteTo / ReadFrom move
 * (ecc/bn254/marshal.go:801-823, :907-948; G2 :1051-1075, :1168-1215; bls12-381/marshal.go:25-35 for the 3-bit flags): X only,
 * gmsm_affine_limbs(group) * 4 bytes a point (Fp2: X.A1 | X.A0), the flag bits say which root Y is (LexicographicallyLargest).
 *   gmsm_points_from_compressed: decode (Y = sqrt(X^3 + b) on the device: the compute-heavy half of Decoder.Decode,
 *      unsafeComputeY, marshal.go:951-989) + the checks of `check` (the curve equation holds by construction; 2 / 3 add the
 *      subgroup test).  Errors as the reference words them: "invalid compressed coordinate: square root doesn't exist",
 *      "invalid infinity point encoding", "invalid fp.Element encoding", "invalid point: subgroup check failed"; uncompressed
 *      or undefined flags are refused (gmsm_points_from_raw takes those).
 *   gmsm_points_compress: the encoder, points (host) or d_points (device) -> out_comp (host).
 *   gmsm_bases_register_compressed: decode + check + register in one call. */
int gmsm_points_from_compressed(int group, const uint8_t *comp, size_t n, int check, uint64_t *out_affine, void *d_out_affine,
                                int64_t *bad_index);
int gmsm_points_compress(int group, const uint64_t *points, const void *d_points, size_t n, uint8_t *out_comp);
int gmsm_bases_register_compressed(int group, const uint8_t *comp, size_t n, int check, uint64_t *out_handle, int64_t *bad_index);
int gmsm_bases_register_dump(int group, const char *path, uint64_t offset, int expect_marker, size_t max_points, int check,
                             uint64_t *out_handle, size_t *out_n, int64_t *bad_index);

/* ---- fr/fft: ecc/<curve>/fr/fft on the device, over the scalar field of `group`'s curve (G1 and
 *      G2 ids of a curve name the same field).  gmsm_fft_domain_new = fft.NewDomain(m) (fr/fft/domain.go:66): cardinality
 *      = next power of two >= m, Generator = fr.Generator(m) (fr/generator.go:18-36), coset shift = the generator of Fr^*
 *      (domain.go:56-62; 5 / 7 / 15 for BN254 / BLS12-381 / BW6-761); twiddles live in HBM, coset tables appear with the
 *      first coset transform.  gmsm_fft = (*Domain).FFT (inverse = 0) / FFTInverse (inverse != 0), fft.go:31-196:
 *      decimation 0 = DIT (input bit-reversed, output natural), 1 = DIF (input natural, output bit-reversed);
 *      on_coset = the OnCoset() option.  a: n = cardinality fr.Elements (Montgomery limbs), transformed in place, either on
 *      the host (`a`) or on the device (`d_a`, produced on hip_stream; the call returns when the result is complete).
 *      gmsm_fft_bit_reverse = fft.BitReverse (bitreverse.go:20).  Results are bit-identical to the reference's: every
 *      output is a uniquely determined field element in canonical Montgomery form. ---- */
int gmsm_fft_domain_new(int group, uint64_t m, uint64_t *out_handle);
int gmsm_fft_domain_release(uint64_t handle);
/* Cardinality and the domain's field constants (fr.Element limbs each; any pointer may be NULL) */
int gmsm_fft_domain_info(uint64_t handle, uint64_t *cardinality, uint64_t *generator, uint64_t *generator_inv,
                         uint64_t *cardinality_inv, uint64_t *fr_multiplicative_gen, uint64_t *fr_multiplicative_gen_inv);
int gmsm_fft(uint64_t handle, uint64_t *a, void *d_a, size_t n, int inverse, int decimation, int on_coset, void *hip_stream);
int gmsm_fft_bit_reverse(int group, uint64_t *a, void *d_a, size_t n, void *hip_stream);

/* ---- window-sharded pieces (multi-GPU: windows win_first, win_first+win_stride, ... of the c-bit decomposition are
 *      handled by this device; the tiny per-window totals are exchanged by the caller, e.g. one RCCL all-gather).
 *      out_xyzz (host) receives nwin_local x {X,Y,ZZ,ZZZ} extended-Jacobian window totals
 *      (= what processChunkG1Jacobian sends on chRes, multiexp_jacobian.go:60). ---- */
unsigned gmsm_default_window_bits(int group, size_t n);            /* the engine's choice of c for n points: a measured
                                                                      table per group, 8..17; c = 17 (BN254 G1 from 2^22
                                                                      points) is beyond the reference's uint16 digits */
unsigned gmsm_num_windows(int group, unsigned c);                  /* computeNbChunks, multiexp.go:681 */
/* What a MultiExp over n bases taken anew runs as under the current options: window width, number of windows,
 * entries_per_point (2 = GLV half scalars: 2 n entries per window, gmsm_glv.h; 1 = full scalars) and fused (1 = the fused
 * small-n kernel, 0 = the sorted pipeline). Cost only - the result does not depend on any of it. NULL pointers are skipped. */
int gmsm_default_plan(int group, size_t n, unsigned *c, unsigned *nwin, unsigned *entries_per_point, unsigned *fused);
/* c: 2..20 (the affine result does not depend on it, multiexp_test.go:95-126; the reference stops at 16); one call takes
 * up to 2^27 points for c <= 17 and 2^(44-c) beyond (32-bit sort entries) - larger inputs: split by point range and add
 * the sets with gmsm_fold_window_sets */
int gmsm_window_sums_device(int group, const void *d_points, const void *d_scalars, size_t n, unsigned c,
                            unsigned win_first, unsigned win_stride, void *hip_stream, uint64_t *out_xyzz);
/* The same pipeline without the copy-back: the nwin_local totals are written to the DEVICE buffer d_out_xyzz in stream
 * order and the call returns without waiting (the next step is an RCCL all-gather on the same stream; here hip_stream
 * NULL means the device's default stream, not an engine-private one). Bases: d_points
 * (Go layout) or, when bases_handle != 0, the registered bases (d_points ignored). */
int gmsm_window_sums_enqueue(int group, const void *d_points, uint64_t bases_handle, const void *d_scalars, size_t n,
                             unsigned c, unsigned win_first, unsigned win_stride, void *hip_stream, void *d_out_xyzz);
/* Point-sharded variant of the fold: xyzz_sets = nsets x nwin window totals, one set per point slice (every slice
 * decomposed with the same c). Window w = sum over the sets (g1JacExtended.add, g1.go:736), then the Horner fold. */
int gmsm_fold_window_sets(int group, unsigned c, const uint64_t *xyzz_sets, unsigned nsets, uint64_t *out_jac);
/* Horner fold of all nwin = gmsm_num_windows(group,c) window totals (msmReduceChunk, multiexp.go:302-315) -> Jacobian */
int gmsm_fold_windows(int group, unsigned c, const uint64_t *xyzz_windows, uint64_t *out_jac);
/* FromJacobian (g1.go:150-166) */
int gmsm_jac_to_affine(int group, const uint64_t *jac, uint64_t *out_affine);

/* ---- introspection / test hooks (used by tests/ to check each stage against the oracle) ---- */
size_t gmsm_affine_limbs(int group);   /* uint64 limbs of one affine point */
size_t gmsm_scalar_limbs(int group);
/* digits[nwin][n] (uint32 codes: 0 skip, d>0 -> 2d, d<0 -> 2(-d-1)+1) for host scalars; test hook for k_decompose */
int gmsm_debug_decompose(int group, const uint64_t *scalars, size_t n, unsigned c, uint32_t *out_digits);
/* test hook for the GLV split of gmsm_glv.h: out = n x 2 x (1 + half_words) uint32 - for scalar i (Montgomery fr limbs, as
 * everywhere) {sign of k1, |k1| words, sign of k2, |k2| words}, half_words = 4 (BN254), 5 (BLS12-381), 6 (BW6-761);
 * k1 + k2 lambda = s mod r (lambdaGLV, ecc/bn254/bn254.go:133) */
int gmsm_debug_glv_split(int group, const uint64_t *scalars, size_t n, uint32_t *out);
/* element-wise field ops on device: op 0 mul, 1 add, 2 sub, 3 neg, 4 dbl, 5 sqr, 6 from_mont; field 0 = fp, 1 = fr,
 * 2 = the group's coordinate field (Fp2 for G2 where applicable), 3 = the same coordinate field computed by the lazy-limb
 * code the pipeline uses (convert in, operate, convert out).  a,b,out: count x limbs host arrays. */
int gmsm_debug_field_op(int group, int field, int op, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out);
/* device group-law hook: out[i] = XYZZ accumulation of  acc[i] (+/-) pts[i]  (op 0 add_mixed, 1 sub_mixed),
 * op 2: out[i] = acc[i] + acc2[i] (xyzz add), op 3: out[i] = 2*acc[i]; ops 4..7: the same four through the lazy-limb
 * group law of the pipeline. */
int gmsm_debug_group_op(int group, int op, const uint64_t *acc, const uint64_t *pts_or_acc2, size_t count, uint64_t *out);

/* ---- utilities ---- */
/* out_points[i] = [k0 + i*k1] * base, i < n (affine, Go layout). k0,k1: plain (non-Montgomery) little-endian limbs.
 * Host-side, nthreads worker threads. For building SRS-like synthetic bases (cf. BatchScalarMultiplicationG1,
 * ecc/bn254/g1.go:1039). */
int gmsm_generate_points(int group, const uint64_t *base_affine, const uint64_t *k0, const uint64_t *k1, int klimbs,
                         size_t n, int nthreads, uint64_t *out_points);
/* Per-stage device timing with HIP events on the launch streams. gmsm_set_profiling(1) resets and enables the
 * accumulators; gmsm_get_stage_times copies the summed milliseconds of up to max_stages stages
 * (0 base rewrite + decompose, 1 histogram, 2 scans, 3 scatter + fine sort, 4 bucket accumulation kernel, 5 split-bucket
 * fixup, 6 bucket reduction, 7 reserved = 0) and the number of pipeline runs they cover; returns the number of stages
 * written. gmsm_get_stage_launches gives how many stage instances each sum covers - for stage 4 that is the number of
 * k_accumulate_seg launches.  gmsm_set_profiling(2): only the accumulation kernel is bracketed (two events per pipeline
 * run instead of eight - the events themselves cost 0.03-0.05 ms per call): stage 4 and its launch count are filled,
 * the other stages stay 0. */
void gmsm_set_profiling(int on);
int gmsm_get_stage_times(double *out_ms, int max_stages, unsigned long *out_calls);
int gmsm_get_stage_launches(unsigned long *out_launches, int max_stages);

/* ---- switches (process-wide).  GMSM_OPT_WINDOW_BITS and GMSM_OPT_TABLES take their initial value from the environment
 *      variables GMSM_C / GMSM_TABLES once, when the library is first used; nothing reads the environment per call.
 *      GMSM_OPT_MAX_RUN / GMSM_OPT_HOST_RANGES exist for the tests of the point-range splits (0 = off). ---- */
enum gmsm_option {
    GMSM_OPT_WINDOW_BITS = 0, /* 0 = the library's measured table per group and size, 2..20 = forced (cost only: the
                                 affine result does not depend on c, multiexp_test.go:95-126) */
    GMSM_OPT_TABLES = 1,      /* window tables of registered bases: 0 never, 1 (default) the measured call sizes, 2 always */
    GMSM_OPT_MAX_RUN = 2,     /* lower the 2^27-point cap of one pipeline run: larger calls split into point ranges */
    GMSM_OPT_HOST_RANGES = 3, /* force the number of point ranges a host-buffer call is cut into */
    GMSM_OPT_FIXED_BASE_BITS = 4, /* table width of gmsm_batch_scalar_mul*: 0 = by batch size (8, and 11 from 2^21 scalars), 2..14 */
    GMSM_OPT_SPIN_WAIT_US = 5, /* a blocking MultiExp polls its stream for this many microseconds before it parks on it
                                 (default 0 = park at once; polling was measured at 5-6 us per call, 0.3-1 %, for a
                                 core kept busy as long as the call runs) */
    GMSM_OPT_SMALL_BITS = 6,  /* the fused small-n kernel (one launch, LDS buckets; calls of at most a few thousand points
                                 that neither force a window width nor are served by window tables): 0 (default) = on, width
                                 by size; 1 = off (the sorted pipeline for every size); 2..7 = on with this window width */
    GMSM_OPT_SMALL_MAX = 7,   /* largest call the fused small-n kernel takes (0 = the measured default) */
    GMSM_OPT_SPLIT = 8,       /* experiment (default 0 = off): a call's windows in two groups, the fix-up + reduction of the first on a
                                 second stream beside the accumulation of the second (measured: profiles/r05_split_groups.log) */
    GMSM_OPT_GLV = 9,         /* GLV half scalars (ecc/utils.go:62-170; s P = k1 P + k2 phi(P), half the windows and half the host
                                 fold for the same group element): 0 never; 1 (default) in the fused small-n kernel and, for
                                 bases taken anew, in the sorted pipeline at the sizes where it was measured ahead (2^13..2^20
                                 by group; nothing when a width is forced); 2 in every unregistered call (A/B).
                                 PRECONDITION while it is on - the one the reference's own mulGLV has (ecc/bn254/g1.go:536-600):
                                 phi(P) = [lambda]P holds on the r-torsion only. A point ON the curve but OUTSIDE the subgroup
                                 (possible where the cofactor is not 1: every group but BN254 G1; never a point that passed
                                 IsInSubGroup / the Decoder's default checks / gmsm_points_validate level 2) then contributes
                                 k1 P + k2 phi(P) instead of s P. The reference's MultiExp never uses the endomorphism and is
                                 the integer combination on ANY curve point: GMSM_OPT_GLV = 0 is exactly that
                                 (tests/test_gpu_glv.py::test_glv_off_is_the_integer_combination_outside_the_subgroup) */
    GMSM_OPT_SMALL_QUAD = 10  /* bucket phase of the fused small-n kernel on lane quads: 0 (default) by call size, 1 never,
                                 2 always (the Fp2 groups and BW6-761 always run it on quads) */
};
int gmsm_set_option(int key, unsigned value);
unsigned gmsm_get_option(int key);

/* ---- lifecycle.  Scratch buffers are grow-only per workspace (a steady stream of calls never allocates); the reference's
 *      per-call buffers are garbage-collected (ecc/bn254/multiexp.go:148-176), so a long-lived process needs the
 *      equivalent:  gmsm_trim releases every scratch / staging buffer larger than keep_bytes of every workspace that is
 *      idle right now, on every device (*out_freed = device bytes given back; registered bases, window tables and FFT
 *      domains stay).  gmsm_shutdown releases everything - handles become unknown, contexts, streams and events are
 *      destroyed - and leaves the library usable (state reappears on first use); no other call may run or start meanwhile,
 *      and it refuses (GMSM_ERR_ARG) while a submitted ticket is uncollected. ---- */
int gmsm_trim(size_t keep_bytes, size_t *out_freed);
int gmsm_shutdown(void);

int gmsm_device_count(void);
/* Device used by later calls of this thread AND process-wide default for threads that never called it (a goroutine
 * that is moved to another OS thread keeps its device as long as the process uses one device; a process that drives
 * several devices from several threads must pin them, runtime.LockOSThread). Entries that take device pointers or a
 * bases handle do not depend on it: they run on the device that owns the pointer / the registered bases.
 * Calling it also pins the drop-in entries to ONE device even when GMSM_DEVICES lists several
 * (gmsm_set_devices(NULL, 0) undoes that). */
int gmsm_set_device(int device);
/* text of the calling thread's last failure; a thread that never failed gets the most recent failure of the process
 * (a cgo caller may be rescheduled onto another OS thread between the failing call and this one) */
const char *gmsm_last_error(void);
const char *gmsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GMSM_H */
