// One (curve, group) instance of the engine: the pipeline driver (struct Group) over the host-side arithmetic of
// gmsm_group_host.h. The test hooks live in gmsm_group_debug.h, the function table the C ABI dispatches through in
// gmsm_group_vtable.h (both included at the end). Instantiated once per group in gmsm_group_inst.hip (one translation
// unit per group so the six groups compile in parallel).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "gmsm_context.h"
#include "gmsm_group_host.h"
#include "gmsm_kernels.h"
#include "gmsm_fixup_q.h"
#include "gmsm_small.h"
#include "gmsm_fixedbase.h"
#include "gmsm_ingest.h"
#include "gmsm_decompress.h"
#include "gmsm_fft.h"

namespace gmsm {

// Tuning constants of the pipeline geometry. In the shipped library GMSM_TUNE(NAME, default) IS its default, folded at
// compile time - none of these is a run-time knob. An experiments build (-DGMSM_EXPERIMENTS, tools/build_ab.sh) reads
// GMSM_<NAME> from the environment instead, which is how the defaults below were swept:
//   LOG2L, REDUCE_LEVELS   buckets per serial reduction thread / two or three reduction levels (0 = the cost model)
//   SEG, SEGMAX            entries per accumulation thread (0 = chosen per launch) and its cap
//   PART_LOG2, STAGE_CAP   coarse partition population, staging slots of the fine sort
//   DEVICE_RANGES          point ranges of a device-resident call beyond the pipeline-run cap
#define GMSM_TUNE(NAME, DFLT) tune_uint("GMSM_" #NAME, (DFLT))

// ------------------------------------------------------------------ one (curve, group)
template <class T> struct LazyOf;
template <class P> struct LazyOf<Fp<P>> { using type = FpU<P>; };
template <class P> struct LazyOf<Fp2<P>> { using type = Fp2U<P>; };

template <class F_, class FrP_, class Consts_, bool NEEDS_TORSION_>
struct Group : GroupHost<F_, FrP_> {
    using Host = GroupHost<F_, FrP_>;  // fold, fold_sets, generate_points, build_fixed_base_table, fold_powers
    using Host::batch_to_affine;
    using Host::build_fixed_base_table;
    using Host::fold;
    using Host::fold_powers;
    using Host::fold_sets;
    using Host::generate_points;
    using Host::scalar_mul;
    using F = F_;
    using FrP = FrP_;
    using Consts = Consts_;                                   // curve coefficient, wire-format flag bits (gmsm_params32.h)
    static constexpr bool NEEDS_TORSION = NEEDS_TORSION_;     // cofactor != 1: subgroup check beyond the curve equation
    using Aff = Affine<F>;
    using Ext = XYZZ<F>;
    using J = Jac<F>;
    static constexpr unsigned FR_BITS = FrP::BITS;
    static constexpr size_t AFF_BYTES = sizeof(Aff);
    static constexpr size_t SCALAR_BYTES = sizeof(Fp<FrP>);
    // unsaturated-limb accumulation (gmsm_fieldu.h) for groups whose coordinates live in Fp; Fp2 groups use the generic
    // saturated kernel
    using U = typename LazyOf<F>::type;  // lazy element type: FpU<P> for Fp coordinates, Fp2U<P> for Fp2 coordinates
    // Group operations are inlined where a kernel has ONE call site of the addition (k_accumulate_seg, k_fixup_seg,
    // k_reduce_serial: a single inlined Fp2 or BW6-761 addition is 60-350 KB of code); everything that combines few
    // elements (long chains, the reduction's combine and level 2) runs on lane quads with the operands in LDS (gmsm_quad.h).
    // fixed-base / normalisation / table-doubling kernels: every type inlines its group operations since the end of round 6 (the wide
    // types called theirs out of line through rounds 2-6 for code size: same-box A/B profiles/r06_fixed_base_inline_ab.log - BN254 G2
    // 2^20 17.7 -> 8.7 ms, BLS12-381 G2 34.4 -> 18.2, BW6-761 53.6 -> 36.1). -DGMSM_INLINE_OPS_BYTES=56 gives the old form back.
#ifndef GMSM_INLINE_OPS_BYTES
#define GMSM_INLINE_OPS_BYTES (28 * 4)
#endif
    static constexpr bool INLINE_OPS = sizeof(U) <= GMSM_INLINE_OPS_BYTES;
    using OpsSerial = UnsatOps<U>;
    using OpsElem = typename OpsSerial::Elem;
    // Level 1 of the bucket reduction = k_reduce_serial (one thread per L buckets, no LDS) + k_combine_q (COMBINE_N
    // (S, W) pairs per workgroup of 4 * COMBINE_N threads); level 2 = k_reduce2_q, at most RED2_TPB level-1 results per window.
#ifndef GMSM_FORK_CONVERT_LOG2
#define GMSM_FORK_CONVERT_LOG2 18
#endif
    static constexpr size_t FORK_CONVERT_MIN = (size_t)1 << GMSM_FORK_CONVERT_LOG2;  // unregistered calls from here on rewrite their bases on a side stream
    static constexpr int COMBINE_N = 64;
    static constexpr int RED2_TPB = 64;
    // The serial part itself runs on quads (k_reduce_serial_q) for every element type but the 9-limb prime field.
    // Measured reduce times one-lane / quad serial (same box): BW6-761 2^20 1.85 / 1.13 ms, BLS12-381 G2 2^22 2.17 / 1.61,
    // BN254 G2 2^20 0.955 / 0.906, BLS12-381 G1 2^22 0.675 / 0.601 (2^16: 0.688 / 0.611).
#ifndef GMSM_SERIAL_QUAD_WORDS
#define GMSM_SERIAL_QUAD_WORDS 14
#endif
    static constexpr bool SERIAL_QUAD = sizeof(U) >= GMSM_SERIAL_QUAD_WORDS * 4;
#ifndef GMSM_SERIAL_Q_QUADS
#define GMSM_SERIAL_Q_QUADS 64
#endif
    static constexpr int SERIAL_Q_QUADS = GMSM_SERIAL_Q_QUADS;  // quads per workgroup of k_reduce_serial_q (A/B: tools/build_ab.sh)
    // The work-efficient combine (k_combine_we: half the issue work, one step more) where the combine is throughput-bound:
    // the 9-limb field with at least two workgroups per CU. Measured k_combine_q / k_combine_we reduce times, BN254 G1:
    // 2^16 0.293 / 0.277 ms, 2^18 0.337 / 0.308, 2^20 0.344 / 0.312, 2^24 0.465 / 0.433 - but 2^12 (a handful of
    // workgroups: latency) 0.080 / 0.106, and the 14-limb field 0.628 / 0.68 (three workgroups per CU, not four).
#ifndef GMSM_COMBINE_WE_WORDS
#define GMSM_COMBINE_WE_WORDS 9
#endif
    static constexpr bool COMBINE_WE = sizeof(U) <= GMSM_COMBINE_WE_WORDS * 4;
    static bool use_combine_we(const Context &ctx, uint32_t workgroups) { return COMBINE_WE && workgroups >= 2u * (uint32_t)ctx.num_cus; }
    // k_fixup_seg: followers a chain head adds itself (one-lane additions); longer chains go to k_fixup_long (quads, one
    // workgroup per chain). Handing chains of 3+ links to the quads for the wide types was measured: no gain at 2^20
    // (BW6-761 fixup 0.75 ms either way - it is 322 K two-link chains in five rounds of 512-register workgroups), and 2^16
    // got slower (0.68 -> 2.6 ms: thousands of short chains, one workgroup each).
    static constexpr uint32_t FIX_MAXWALK = FIXUP_MAXWALK;
    // k_fixup_seg on lane quads (gmsm_fixup_q.h) for the element types whose one-lane addition spills: everything wider than
    // the 14-limb prime field (BN254 G2, BLS12-381 G2, BW6-761). -DGMSM_FIXUP_QUAD_WORDS=1000 builds the one-lane form (A/B).
#ifndef GMSM_FIXUP_QUAD_WORDS
#define GMSM_FIXUP_QUAD_WORDS 15
#endif
    static constexpr bool FIXUP_QUAD = sizeof(U) >= GMSM_FIXUP_QUAD_WORDS * 4;
    static constexpr bool SKEWED_HOST_RANGES = AFF_BYTES <= 96;  // host_ranges / window_sums_from_host
    // plan_geometry: combine workgroups resident on a CU (measured on the pieces of a window-sharded call and on single
    // bucket sets, profiles/r03_reduce_levels.log: with one per CU the model kept few windows on the two-level form)
    static constexpr size_t COMBINE_RESIDENT = sizeof(U) <= 36 ? 4 : sizeof(U) <= 56 ? 3 : sizeof(U) <= 72 ? 2 : 1;

    static WindowPlan make_plan(unsigned c, unsigned win_first, unsigned win_stride) {
        WindowPlan p;
        p.c = c;
        p.nwin_total = num_windows(FR_BITS, c);
        unsigned lc = last_c(FR_BITS, c);
        p.nbuckets = 1u << (std::max(c, lc) - 1);
        p.win_first = win_first;
        p.win_stride = win_stride ? win_stride : 1;
        p.nwin_local = win_first < p.nwin_total ? (p.nwin_total - win_first + p.win_stride - 1) / p.win_stride : 0;
        p.shared = 0;
        p.glv = 0;
        return p;
    }
    // the windows of the GLV half scalars (gmsm_glv.h): same rules over GLV_BITS bits
    static constexpr unsigned GLV_BITS = (unsigned)FrP::GLV_BITS;
    static WindowPlan make_plan_glv(unsigned c, unsigned win_first, unsigned win_stride) {
        WindowPlan p = make_plan(c, win_first, win_stride);
        p.nwin_total = num_windows(GLV_BITS, c);
        p.nbuckets = 1u << (std::max(c, last_c(GLV_BITS, c)) - 1);
        p.nwin_local = win_first < p.nwin_total ? (p.nwin_total - win_first + p.win_stride - 1) / p.win_stride : 0;
        p.glv = 1;
        return p;
    }

    // Runs the device pipeline for the windows of `plan`; host_xyzz receives plan.nwin_local window totals.
    // d_points: Go-layout affine bases on the device, or nullptr when `resident` (bases already rewritten into the lazy
    // domain by register_bases) is given.
    // `ws` is leased by the caller; the pipeline runs on the workspace's own stream, ordered after `caller_stream`.
    static int window_sums(Context &ctx, Workspace &ws, const void *d_points, const void *d_scalars, size_t n,
                           const WindowPlan &plan, hipStream_t caller_stream, Ext *host_xyzz,
                           const ResidentBases *resident = nullptr, size_t resident_offset = 0) {
        int rc = order_after(ws, caller_stream);
        if (rc) return rc;
        if ((rc = enqueue_window_sums(ctx, ws, d_points, d_scalars, n, plan, ws.stream, resident, nullptr, resident_offset)))
            return rc;
        return collect_window_sums(ws, ws.stream, plan.nwin_local, host_xyzz);
    }

    // Waits for the pipeline enqueued on `stream` and hands out the window totals (pinned buffer -> host_xyzz).
    static int collect_window_sums(Workspace &ws, hipStream_t stream, uint32_t nw, Ext *host_xyzz) {
        if (nw == 0) return GMSM_OK;
        HIP_TRY(wait_stream(stream));
        if (ws.pending_timed) StageTimer::collect(ws);
        ws.pending_timed = false;
        memcpy(host_xyzz, ws.pinned, (size_t)nw * sizeof(Ext));
        return GMSM_OK;
    }

    // floor(r / 2^shift) for the scalar-field modulus r (the largest value a top window can hold, before the carry)
    static uint64_t fr_modulus_shifted(unsigned shift) {
        uint64_t v = 0;
        for (int b = 62; b >= 0; --b) {
            const unsigned bit = shift + (unsigned)b;
            if (bit < 32u * FrP::N && ((FrP::Q[bit >> 5] >> (bit & 31)) & 1u)) v |= (uint64_t)1 << b;
        }
        return v;
    }
    // largest digit code k_decompose can emit for this plan (multiexp.go:779-800): decides uint16 vs uint32 digit arrays
    static uint64_t max_digit_code(const WindowPlan &plan) {
        if (plan.glv) {  // half scalars below 2^GLV_BITS: the top window holds at most 2^(its bits) (value + carry), code = 2 digit
            const uint64_t low = ((uint64_t)1 << plan.c) - 1;
            const unsigned top_bits = GLV_BITS - (plan.nwin_total - 1) * plan.c;
            return std::max(low, (uint64_t)2 << top_bits);
        }
        const uint64_t low = ((uint64_t)1 << plan.c) - 1;
        const unsigned top_shift = (plan.nwin_total - 1) * plan.c;
        const uint64_t top = top_shift >= 63 + 32u * FrP::N ? 0 : ((fr_modulus_shifted(top_shift) + 1) << 1);
        return plan.nwin_total > 1 ? std::max(low, top) : top;
    }

    // Accumulation and reduction geometry of one pipeline run over nw windows of n points.
    struct Geometry {
        uint32_t seg, tpw;                     // accumulation: entries per thread, threads per window
        uint32_t log2L, nblocks1, log2span;    // reduction: buckets per serial thread, level-1 workgroups per window
        uint32_t nblocks2;                     // 0: serial - combine - level 2; else a second combine of nblocks2 workgroups
    };

    static Geometry plan_geometry(const Context &ctx, uint32_t nw, size_t n, uint32_t NB, bool shared_set = false) {
        Geometry q;
        // reduction: buckets per serial thread, L = 2^log2L; COMBINE_N of their (S, W) pairs per combine workgroup; the
        // last level (k_reduce2_q) takes at most RED2_TPB blocks per window, one quad each. Few windows of many buckets
        // (one rank's share of a window-sharded call, the single bucket set of the window tables) would need a long
        // serial walk to get there: a second combine level in between keeps L short.
        constexpr size_t SPAN1 = COMBINE_N;  // pairs one combine workgroup takes
        const auto blocks1 = [&](uint32_t l2) { return ((size_t)NB + (SPAN1 << l2) - 1) / (SPAN1 << l2); };
        const auto blocks2 = [&](uint32_t l2) { return (blocks1(l2) + SPAN1 - 1) / SPAN1; };
        uint32_t log2L = GMSM_TUNE(LOG2L, 0);
        const uint32_t force_levels = GMSM_TUNE(REDUCE_LEVELS, 0);  // 2 / 3; 0 = cost model
        bool three = force_levels == 3;
        if (log2L == 0) {
            // serial kernel: 2L dependent one-lane additions on every SIMD, as many rounds as the threads need;
            // combine: 2 log2 N + log2 L + 1 quad steps (a quad step is about a third of a one-lane addition) per
            // round of workgroups. Minimise the total in units of one-lane additions.
            // The width of the TWO-level form is chosen with one combine workgroup per CU and round - the model all
            // full-call configurations were measured with. Whether a second combine level pays is then decided with
            // the workgroups a CU really holds at once (COMBINE_RESIDENT; their steps interleave): with one per CU the
            // model kept few windows on a long serial walk (2^20 piece of 8: reduce 0.221 ms, 0.176 with three levels;
            // 2^24 piece of 4: 0.358 / 0.271 - profiles/r03_reduce_levels.log).
            const auto costs = [&](uint32_t l2, size_t resident, size_t *c3) {
                const size_t serial_threads = (size_t)nw * (((size_t)NB + ((size_t)1 << l2) - 1) >> l2);
                // lanes of the serial kernel: one per thread, or a quad per thread at a third of the step time
                const size_t serial_lanes = SERIAL_QUAD ? 4 * serial_threads : serial_threads;
                const size_t serial_rounds = (serial_lanes + (size_t)ctx.num_cus * 256 - 1) / ((size_t)ctx.num_cus * 256);
                const size_t per_round = (size_t)ctx.num_cus * resident;
                const size_t rounds = ((size_t)nw * blocks1(l2) + per_round - 1) / per_round;
                const size_t cost_serial = (SERIAL_QUAD ? 1 : 3) * serial_rounds * ((size_t)2 << l2);
                const size_t c2 = cost_serial + rounds * (2 * 6 + l2 + 1);
                // second combine: its own rounds of 2 * 6 + 1 steps + the tail of the prescaling, and one more launch
                const size_t rounds_b = ((size_t)nw * blocks2(l2) + per_round - 1) / per_round;
                *c3 = c2 + rounds_b * (2 * 6 + 8) + 4;
                return c2;
            };
            size_t best2 = ~(size_t)0, best3 = ~(size_t)0, unused = 0;
            uint32_t l2_two = 0, l2_three = 0;
            for (uint32_t l2 = 1; l2 <= 8; ++l2) {
                const size_t c2 = costs(l2, 1, &unused);
                if (blocks1(l2) <= (size_t)RED2_TPB && c2 < best2) {
                    best2 = c2;
                    l2_two = l2;
                }
                size_t c3 = 0;
                (void)costs(l2, COMBINE_RESIDENT, &c3);
                if (blocks1(l2) > 1 && blocks2(l2) <= (size_t)RED2_TPB && c3 < best3) {
                    best3 = c3;
                    l2_three = l2;
                }
            }
            const size_t two_resident = l2_two ? costs(l2_two, COMBINE_RESIDENT, &unused) : ~(size_t)0;
            if (force_levels == 3 ? l2_three != 0 : (force_levels != 2 && l2_three != 0 && best3 < two_resident)) {
                log2L = l2_three;
                three = true;
            } else if (l2_two) {
                log2L = l2_two;
                three = false;
            } else {
                log2L = 8;
            }
        }
        if (three) {
            while (blocks2(log2L) > (size_t)RED2_TPB) ++log2L;
        } else {
            while (blocks1(log2L) > (size_t)RED2_TPB) ++log2L;
        }
        q.log2L = log2L;
        q.nblocks1 = (uint32_t)blocks1(log2L);
        q.nblocks2 = three ? (uint32_t)blocks2(log2L) : 0u;
        q.log2span = log2L;
        for (size_t t = SPAN1; t > 1; t >>= 1) ++q.log2span;
        // entry-parallel segmented accumulation: seg entries per thread. Every thread does the same work, so the
        // launch should be a whole number of resident "rounds" of WORKGROUPS: the grid is (blocks per window) x
        // (windows), so the unit is a 256-thread block per window - nw * ceil(tpw/256) blocks must not exceed
        // rounds x (CUs x resident blocks). (Counting threads instead put 24 x 11 = 264 blocks on 256 CUs for the
        // 24 windows of BW6-761: eight blocks ran a second round alone and the kernel took twice as long - PMC:
        // 1056 waves, each alive for half of the kernel.) Among the round counts that keep seg <= SEG_MAX the
        // one that fills its last round best is taken.
        uint32_t seg = GMSM_TUNE(SEG, 0);
        if (seg == 0) {
            const size_t cap_blocks = (size_t)ctx.num_cus * AccWaves<U>::value;
            const size_t SEG_MAX = GMSM_TUNE(SEGMAX, 512);  // measured: 512 best at 2^24, 256 at 2^22
            // Shortest segment. A launch that does not fill the chip is a latency chain of `seg` additions per thread:
            // with sparse buckets (<= 8 entries each: few chains of partial sums to close) shorter segments on more
            // threads win - BN254 G1 2^14 0.605 -> 0.476 ms, 2^16 0.637 -> 0.585, BLS12-381 G1 2^16 1.17 -> 0.98; crowded
            // buckets (the 8-bit windows of small inputs) would pay it back in the fix-up (BLS12-381 G1 2^14: 0.13 -> 0.49 ms).
            // Measured for the three narrow element types (profiles/r03_short_segments.log).
            size_t seg_floor = 32;
            if (sizeof(U) <= 72 && n / NB <= 8) {
                const size_t entries = (size_t)nw * n, cap_threads = cap_blocks * 256;
                seg_floor = entries / 8 <= cap_threads ? 8 : entries / 16 <= cap_threads ? 16 : 32;
            } else if (sizeof(U) <= 72 && shared_set) {
                // the shared bucket set of the window tables is crowded by construction, but its chains are closed one
                // thread per bucket (k_fixup_bucket): BN254 G1 2^14 0.456 -> 0.365 ms at 8, 2^16 0.472 -> 0.400 at 16
                // (0.428 at 8), 2^17 and up best at 32; BLS12-381 G1 2^15 0.775 -> 0.601 at 8, 2^16 0.817 -> 0.657 at 16;
                // BN254 G2 (its one-lane additions cost three times as much) 2^15 1.088 -> 0.824 at 16, 2^16 best at 32
                const size_t entries = (size_t)nw * n;
                if (sizeof(U) <= 56) seg_floor = entries <= ((size_t)1 << 19) ? 8 : entries <= ((size_t)3 << 19) ? 16 : 32;
                else seg_floor = entries <= ((size_t)3 << 18) ? 16 : 32;
            }
            size_t best_seg = 0, first_r = 0;
            double best_fill = -1.0;
            for (size_t r = 1; r < 100000; ++r) {
                const size_t bpw = r * cap_blocks / nw;  // whole blocks per window in r rounds
                if (bpw == 0) continue;
                const size_t sg = std::max<size_t>((n + bpw * 256 - 1) / (bpw * 256), seg_floor);
                if (sg > SEG_MAX) continue;
                if (!first_r) first_r = r;
                const size_t blocks = (size_t)nw * (((n + sg - 1) / sg + 255) / 256);
                const double fill = (double)blocks / (double)(((blocks + cap_blocks - 1) / cap_blocks) * cap_blocks);
                if (fill > best_fill + 1e-9) {
                    best_fill = fill;
                    best_seg = sg;
                }
                if (fill > 0.985 || r >= first_r + 5 || sg == seg_floor) break;
            }
            seg = (uint32_t)(best_seg ? best_seg : seg_floor);
        }
        q.seg = seg;
        q.tpw = (uint32_t)((n + seg - 1) / seg);  // threads per window (upper bound: <= n entries)
        return q;
    }

    // k_decompose for this plan: the unrolled kernel of the plan's width when it is one of the window table's
    // (preferred_c), the generic one otherwise (forced and test widths). d16: uint16 digit codes.
    static void launch_decompose(const void *d_scalars, size_t n, const WindowPlan &plan, bool d16, void *digits,
                                 const uint8_t *skip, hipStream_t stream) {
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (plan.glv) {  // half scalars: digits[k][2 n] (glv_width_built() lists the widths)
#define GMSM_DECOMPOSE_GLV(CC)                                                                                                  \
    case CC:                                                                                                                    \
        if (d16) hipLaunchKernelGGL((k_decompose_glv<FrP, uint16_t, CC>), grid, block, 0, stream, (const uint32_t *)d_scalars, n, \
                                    plan, (uint16_t *)digits, skip);                                                             \
        else hipLaunchKernelGGL((k_decompose_glv<FrP, uint32_t, CC>), grid, block, 0, stream, (const uint32_t *)d_scalars, n,     \
                                plan, (uint32_t *)digits, skip);                                                                 \
        return;
            switch (plan.c) {
                GMSM_DECOMPOSE_GLV(10) GMSM_DECOMPOSE_GLV(11) GMSM_DECOMPOSE_GLV(12) GMSM_DECOMPOSE_GLV(13) GMSM_DECOMPOSE_GLV(14)
                GMSM_DECOMPOSE_GLV(15) GMSM_DECOMPOSE_GLV(16) GMSM_DECOMPOSE_GLV(17) GMSM_DECOMPOSE_GLV(18) GMSM_DECOMPOSE_GLV(20)
                default: return;  // never planned (glv_width_built)
            }
#undef GMSM_DECOMPOSE_GLV
        }
#define GMSM_DECOMPOSE_C(CC)                                                                                                \
    case CC:                                                                                                                \
        if (d16) hipLaunchKernelGGL((k_decompose_c<FrP, uint16_t, CC>), grid, block, 0, stream, (const uint32_t *)d_scalars, n,  \
                                    plan, (uint16_t *)digits, skip);                                                         \
        else hipLaunchKernelGGL((k_decompose_c<FrP, uint32_t, CC>), grid, block, 0, stream, (const uint32_t *)d_scalars, n,      \
                                plan, (uint32_t *)digits, skip);                                                             \
        return;
        switch (plan.c) {
            GMSM_DECOMPOSE_C(8) GMSM_DECOMPOSE_C(9) GMSM_DECOMPOSE_C(10) GMSM_DECOMPOSE_C(12) GMSM_DECOMPOSE_C(13)
            GMSM_DECOMPOSE_C(14) GMSM_DECOMPOSE_C(15) GMSM_DECOMPOSE_C(16) GMSM_DECOMPOSE_C(17) GMSM_DECOMPOSE_C(18)
            GMSM_DECOMPOSE_C(20)
            default: break;
        }
#undef GMSM_DECOMPOSE_C
        if (d16)
            hipLaunchKernelGGL((k_decompose<FrP, uint16_t>), grid, block, 0, stream, (const uint32_t *)d_scalars, n, plan,
                               (uint16_t *)digits, skip);
        else
            hipLaunchKernelGGL((k_decompose<FrP, uint32_t>), grid, block, 0, stream, (const uint32_t *)d_scalars, n, plan,
                               (uint32_t *)digits, skip);
    }

    // Launches every kernel of one pipeline run and queues the copy of the window totals into ws.pinned; does not wait.
    // `stream` carries the call (inputs are ready there; the totals are complete there). All scratch comes from `ws`, so
    // two workspaces can be in flight at once.
    // d_out != nullptr: the totals stay on the device (copied to d_out in stream order) instead of going to ws.pinned.
    // (Round 2 could cut the windows of a call into groups on two streams so that the reduction of one group ran under
    // the accumulation of the next; measured slower in every configuration - profiles/r02_split_ab.log - and removed.)
    // buckets_only: stop after the fix-up - ws.buckets / ws.starts hold the range's bucket sums (a range of a multi-range
    // host call: k_merge_buckets adds them to the call's running buckets, ONE reduction at the end; enqueue_reduce).
    static int enqueue_window_sums(Context &ctx, Workspace &ws, const void *d_points, const void *d_scalars, size_t n,
                                   const WindowPlan &plan, hipStream_t stream, const ResidentBases *resident,
                                   void *d_out = nullptr, size_t resident_offset = 0 /* first registered base used */,
                                   bool buckets_only = false, bool time_range = false /* buckets_only: the caller collects the
                                   stage events of this range before it enqueues the next (multiexp_device) */) {
        const uint32_t nwd = plan.nwin_local;  // windows of the digit decomposition = totals handed out
        ws.pending_timed = false;
        if (nwd == 0) return GMSM_OK;
        // window tables: the (window, point) pairs of all windows form ONE set of entries over one bucket set
        const bool shared = plan.shared != 0;
        if (shared && !(resident && resident->tab_c.load() == plan.c && plan.win_first == 0 && plan.win_stride == 1))
            return fail(GMSM_ERR_ARG, "shared-bucket plan without matching window tables");
        if (shared && n) g_table_runs.fetch_add(1, std::memory_order_relaxed);
        if (ws.uncollected) {  // stage events of an enqueue-only call (nobody waited for it): pick them up now
            if (ws.timed && hipEventQuery(ws.events[ws.timed_level == 2 ? T_FIXUP : T_END]) == hipSuccess) StageTimer::collect(ws);
            ws.uncollected = false;
        }
        // The workspace may still be in use by work enqueued earlier on another stream: order behind it.
        int rc;
        if ((rc = begin_use(ws, stream))) return rc;
        if (n == 0) {
            int rc0 = ws.ensure_pinned((size_t)nwd * sizeof(Ext));
            if (rc0) return rc0;
            if (d_out) HIP_TRY(hipStreamSynchronize(stream));  // an earlier copy out of ws.pinned may be in flight
            for (uint32_t k = 0; k < nwd; ++k) ((Ext *)ws.pinned)[k] = Ext::infinity();
            if (d_out) {
                HIP_TRY(hipMemcpyAsync(d_out, ws.pinned, (size_t)nwd * sizeof(Ext), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            return GMSM_OK;
        }
        const size_t n_points = n;  // scalars / bases of this run
        const uint32_t nw = shared ? 1u : nwd;  // bucket sets: what the sort, the accumulation and the reduction see as windows
        if (shared) n *= nwd;                   // ... and their entries
        const bool glv = plan.glv != 0;         // half scalars: entry 2 i = (P_i, k1_i), entry 2 i + 1 = (phi(P_i), k2_i)
        if (glv && (shared || resident)) return fail(GMSM_ERR_ARG, "GLV plan over registered bases");
        if (glv) n *= 2;
        if (n >= ((size_t)1 << 31)) return fail(GMSM_ERR_ARG, "n must be < 2^31");
        const uint32_t NB = plan.nbuckets;
        constexpr size_t REC = sizeof(typename OpsSerial::Mem);  // bucket / partial record (lazy representation on the fast path)

        const Geometry q = plan_geometry(ctx, nw, n, NB, shared);
        // GMSM_OPT_SPLIT (experiment): two window groups; each group's accumulation gets the segment length that fills the chip
        // with HALF the windows (the reduction geometry stays the call's)
        const bool split = options().split.load(std::memory_order_relaxed) != 0 && !buckets_only && !shared && nw >= 4 && q.nblocks2 == 0;
        const Geometry qa = split ? plan_geometry(ctx, nw / 2, n, NB, shared) : q;  // accumulation + fix-up geometry
        const size_t tot_thr = (size_t)nw * qa.tpw;
        const size_t tot_blk = (size_t)nw * q.nblocks1;

        // ---- grouping geometry
        uint32_t log2NB = 0;
        while ((1u << log2NB) < NB) ++log2NB;
        uint32_t log2n = 0;
        while (((size_t)1 << log2n) < n) ++log2n;
        const uint32_t lidx = log2n + 1;  // bits of (index << 1 | negate)
        // target partition population 2^part_log2 (measured, BN254 G1: 2^13 is best up to 2^21 points - more
        // workgroups for the fine pass; 2^15 from 2^24 on - 128-byte runs out of the coarse pass)
        const uint32_t part_log2 = GMSM_TUNE(PART_LOG2, log2n <= 21 ? 13 : log2n >= 24 ? 15 : 14);
        int fb = (int)part_log2 + (int)log2NB - (int)log2n;
        if (fb > (int)log2NB) fb = (int)log2NB;
        if (fb > 15) fb = 15;  // the fine pass keeps 2^fb counters in LDS (128 KiB): windows wider than 16 bits on few points
        if (fb < 0) fb = 0;
        if (fb + lidx > 32) fb = 32 - lidx;
        const uint32_t fbits = (uint32_t)fb;
        const uint32_t nparts = NB >> fbits;
        const bool big_chunk = n > ((size_t)1 << 25) && (size_t)nparts * 8 + PART_CHUNK_BIG * 6 <= 152 * 1024;
        const size_t pchunk_len = big_chunk ? PART_CHUNK_BIG : PART_CHUNK;  // chunk = one LDS staging buffer
        const uint32_t pchunks = (uint32_t)((n + pchunk_len - 1) / pchunk_len);
        const size_t scatter_lds = (size_t)nparts * 8 + pchunk_len * 6;
        if (scatter_lds > 152 * 1024)  // 32-bit sort entries: bucket bits + index bits; reached beyond 2^27 points at c <= 17
            return fail(GMSM_ERR_ARG, plan.c > 17 ? "window width too large for this many points in one pipeline run (a run takes up to "
                                                    "2^(44-c) points for c > 17): use a smaller c or split by point range"
                                                  : "more than 2^27 points in one pipeline run: split the window sums by point range "
                                                    "and add the sets (gmsm_fold_window_sets); the MultiExp entries do this themselves");
        // staging slots of the fine pass: up to 96 KiB next to the 2^fbits counters; larger partitions go direct
        const size_t fine_cnt_bytes = (size_t)4 << fbits;
        const uint32_t stage_cap = fine_cnt_bytes >= 156 * 1024 ? 0u
                                   : (uint32_t)std::min<size_t>(GMSM_TUNE(STAGE_CAP, part_log2 >= 15 ? 39000 : 24576),
                                                              (156 * 1024 - fine_cnt_bytes) / 4);
        if (stage_cap > 40u * 1024u) return fail(GMSM_ERR_ARG, "window geometry: staging slots beyond the fine sort's registers");
        if (fine_cnt_bytes + (size_t)stage_cap * 4 > 160 * 1024)  // cannot happen with fb <= 15; a failed launch must not
            return fail(GMSM_ERR_ARG, "window geometry: fine-sort counters exceed the LDS");  // leave garbage for the next kernels
        const bool d16 = max_digit_code(plan) < 65536;
        const size_t dsz = d16 ? 2 : 4;

        // ---- scratch
        if ((rc = ws.digits.ensure((size_t)nw * n * dsz))) return rc;
        if ((rc = ws.sorted.ensure((size_t)nw * n * 4))) return rc;
        if ((rc = ws.parted.ensure((size_t)nw * n * 4))) return rc;
        if ((rc = ws.starts.ensure((size_t)nw * (NB + 1) * 4))) return rc;
        if ((rc = ws.buckets.ensure((size_t)nw * NB * REC))) return rc;
        if ((rc = ws.partials.ensure((tot_blk + (size_t)nw * q.nblocks2) * 2 * REC))) return rc;
        const uint32_t T = (uint32_t)(((size_t)NB + ((size_t)1 << q.log2L) - 1) >> q.log2L);  // (S, W) pairs per window of k_reduce_serial
        if ((rc = ws.red_pre.ensure((size_t)nw * T * 2 * REC))) return rc;
        if ((rc = ws.totals.ensure((size_t)nwd * sizeof(Ext)))) return rc;
        if ((rc = ws.ensure_pinned((size_t)nwd * sizeof(Ext)))) return rc;
        if ((rc = ws.blockhist.ensure((size_t)nw * pchunks * nparts * 4))) return rc;
        if ((rc = ws.counts.ensure((size_t)nw * (2 * nparts + 1) * 4))) return rc;
        if ((rc = ws.seg_partials.ensure(tot_thr * 2 * REC))) return rc;
        if ((rc = ws.seg_flags.ensure(tot_thr * 4))) return rc;
        if ((rc = ws.seg_bucket.ensure(tot_thr * 4))) return rc;
        // long-chain list of the fixup (the fix-up kernels append pieces of LONG_PIECE links, k_fixup_long consumes): one counter
        // (slot of the first window; k_part_rowscan zeroes it) + the items: chains have more than FIX_MAXWALK followers
        const size_t list_cap = 2 * (tot_thr / FIX_MAXWALK + tot_thr / LONG_PIECE + nw + 16);  // two halves: one per window group (GMSM_OPT_SPLIT)
        if ((rc = ws.seg_lvl.ensure((size_t)nw * 4 + 16 + list_cap * sizeof(LongChain)))) return rc;
        uint32_t *long_flag = (uint32_t *)ws.seg_lvl.ptr;                                  // [nw] counters, [0] is used
        LongChain *long_list = (LongChain *)((char *)ws.seg_lvl.ptr + (((size_t)nw * 4 + 15) / 16) * 16);
        const size_t piece_sums_off = ((list_cap * 4 + 15) / 16) * 16;
        if ((rc = ws.long_pieces.ensure(piece_sums_off + list_cap * REC))) return rc;
        uint32_t *piece_done = (uint32_t *)ws.long_pieces.ptr;
        void *piece_sums = (char *)ws.long_pieces.ptr + piece_sums_off;
        // oversized partitions of the fine sort (k_part_rowscan lists them, k_heavy_* sort them): a window has at most
        // n / (stage_cap + 1) of them, and their sub-runs of HEAVY_SUB references number at most n / HEAVY_SUB + that
        const uint32_t hcap = (uint32_t)std::min<size_t>(nparts, n / ((size_t)stage_cap + 1) + 1);
        const uint32_t scap = (uint32_t)(n / HEAVY_SUB) + hcap;
        const size_t hparts_bytes = (((size_t)nw * hcap * sizeof(HeavyPart) + 15) / 16) * 16, hcount_bytes = (((size_t)nw * 8 + 4 + 15) / 16) * 16;
        if ((rc = ws.heavy.ensure(hparts_bytes + hcount_bytes + ((size_t)nw * scap << fbits) * 4))) return rc;
        HeavyPart *hparts = (HeavyPart *)ws.heavy.ptr;
        uint32_t *hcount = (uint32_t *)((char *)ws.heavy.ptr + hparts_bytes);
        uint32_t *subhist = (uint32_t *)((char *)ws.heavy.ptr + hparts_bytes + hcount_bytes);
        if (nw > HEAVY_MAX_WINDOWS) return fail(GMSM_ERR_ARG, "more than 256 windows in one pipeline run");
        // workgroups of the heavy kernels (the sub-runs of all windows are taken round-robin): the chip twice over
        const uint32_t heavy_wg = (uint32_t)std::min<size_t>((size_t)2 * ctx.num_cus, std::max<size_t>((size_t)nw * scap, 1));
        uint32_t *bh = (uint32_t *)ws.blockhist.ptr, *part_base = (uint32_t *)ws.counts.ptr;
        uint32_t *part_pop = part_base + (size_t)nw * (nparts + 1);

        if ((rc = ctx.allow_lds((const void *)k_part_hist<uint16_t>, 160 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_part_hist<uint32_t>, 160 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_part_scatter<uint16_t, PART_CHUNK>, 152 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_part_scatter<uint32_t, PART_CHUNK>, 152 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_part_scatter<uint16_t, PART_CHUNK_BIG>, 152 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_part_scatter<uint32_t, PART_CHUNK_BIG>, 152 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_fine_sort<24>, 160 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_fine_sort<40>, 160 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_heavy_hist, 128 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_heavy_scan, 128 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_heavy_place, 128 * 1024))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_fixup_long<U>, (int)(128 * sizeof(QRec<U>))))) return rc;
        if constexpr (FIXUP_QUAD)
            if ((rc = ctx.allow_lds((const void *)k_fixup_seg_q<U>, (int)(128 * sizeof(QRec<U>))))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_combine_q<U, true, COMBINE_N>, (int)((2 * COMBINE_N + 1) * sizeof(QRec<U>))))) return rc;
        if constexpr (COMBINE_WE)
            if ((rc = ctx.allow_lds((const void *)k_combine_we<U, true>, (int)((2 * COMBINE_N + 1) * sizeof(QRec<U>))))) return rc;
        if constexpr (SERIAL_QUAD)
            if ((rc = ctx.allow_lds((const void *)k_reduce_serial_q<U, SERIAL_Q_QUADS>, (int)(3 * SERIAL_Q_QUADS * sizeof(QRec<U>))))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_reduce2_q<U, true>, (int)(2 * RED2_TPB * sizeof(QRec<U>))))) return rc;

        // Stage times cover single-run calls: the ranges of a multi-range call (buckets_only) reuse a workspace's events
        // before anybody could read them, and their merges and the one reduction run on another stream - such calls are
        // left out of the profile instead of being reported with two of their ranges and no reduction.
        StageTimer timer(ws, /*enabled=*/!buckets_only || time_range);
        ws.timed = timer.on;
        // ---- 0. inputs: rewrite the bases into the lazy Montgomery domain + infinity flags (unless registered earlier),
        // signed-digit decomposition of every scalar
        timer.mark(T_DECOMPOSE, stream);
        const uint8_t *skip = nullptr;
        const void *upoints = nullptr;
        bool forked = false;
        if (resident) {
            upoints = (const char *)(shared ? resident->tables.ptr : resident->upoints.ptr) + resident_offset * AFF_BYTES;
            skip = (const uint8_t *)resident->skip.ptr + resident_offset;
        } else {
            // Round 4: the rewrite of the bases runs on a stream of its own BESIDE the scalar pipeline (decomposition and
            // the two sort passes touch only the scalars; those kernels are bound by LDS atomics and latency, the rewrite
            // by HBM) and joins before the accumulation. Infinity points are recognised by the accumulation from the
            // rewritten record (uaffine_is_infinity), so the decomposition needs no flags from here. 2^20: 36 us off the
            // critical path, 2^24: 0.44 ms. Small calls keep the single stream (two events cost more than they hide).
            if ((rc = ws.upoints.ensure((glv ? 2 : 1) * n_points * AFF_BYTES))) return rc;
            forked = n_points >= FORK_CONVERT_MIN;
            if (forked && (rc = ws.side_stream(ws.cstream))) return rc;
            hipStream_t cs = forked ? ws.cstream : stream;
            if (forked) {
                HIP_TRY(hipEventRecord(ws.ev_fork, stream));
                HIP_TRY(hipStreamWaitEvent(cs, ws.ev_fork, 0));
            }
            if (glv)
                hipLaunchKernelGGL((k_convert_points_glv<U, Consts>), dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                                   cs, d_points, n_points, ws.upoints.ptr);
            else
                hipLaunchKernelGGL((k_convert_points<U>), dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                                   cs, d_points, n_points, ws.upoints.ptr, (uint8_t *)nullptr);
            if (forked) {
                ws.conv_pending = true;  // until the join below is enqueued (an error return in between leaves cstream running)
                HIP_TRY(hipEventRecord(ws.ev_conv, cs));
            }
            upoints = ws.upoints.ptr;
            skip = nullptr;
        }
        // (A decomposition fused with the coarse histogram was measured and dropped: one workgroup per 16 K-scalar chunk
        // leaves 3/4 of the CUs idle at 2^20, and at 2^24 it only breaks even.)
        // (shared: digits[w][i] of the n_points scalars, read from here on as one window of nwd * n_points entries)
        launch_decompose(d_scalars, n_points, plan, d16, ws.digits.ptr, skip, stream);

        const void *digits = ws.digits.ptr;
        uint32_t *sorted = (uint32_t *)ws.sorted.ptr, *parted = (uint32_t *)ws.parted.ptr, *starts = (uint32_t *)ws.starts.ptr;
        char *buckets = (char *)ws.buckets.ptr, *seg_partials = (char *)ws.seg_partials.ptr;
        uint32_t *seg_flags = (uint32_t *)ws.seg_flags.ptr, *seg_bucket = (uint32_t *)ws.seg_bucket.ptr;

        // ---- 1. group the point references of every window by bucket
        timer.mark(T_HIST, stream);
        if (d16)
            hipLaunchKernelGGL(k_part_hist<uint16_t>, dim3(pchunks, nw), dim3(1024), (size_t)nparts * 4, stream,
                               (const uint16_t *)digits, n, nparts, fbits, pchunk_len, bh);
        else
            hipLaunchKernelGGL(k_part_hist<uint32_t>, dim3(pchunks, nw), dim3(1024), (size_t)nparts * 4, stream,
                               (const uint32_t *)digits, n, nparts, fbits, pchunk_len, bh);
        timer.mark(T_SCAN, stream);
        if ((size_t)((nparts + 31) / 32) * nw < (size_t)ctx.num_cus && pchunks >= 256)
            hipLaunchKernelGGL(k_part_colscan<32>, dim3((nparts + 31) / 32, nw), dim3(1024), 0, stream, bh, pchunks, nparts, part_pop, hcount + 2 * nw);
        else
            hipLaunchKernelGGL(k_part_colscan<8>, dim3((nparts + 31) / 32, nw), dim3(256), 0, stream, bh, pchunks, nparts, part_pop, hcount + 2 * nw);
        hipLaunchKernelGGL(k_part_rowscan, dim3(nw), dim3(1024), 0, stream, part_pop, nparts, part_base, long_flag, stage_cap, hparts,
                           hcount, hcap);
        timer.mark(T_SCATTER, stream);
        {
            const dim3 grid(pchunks, nw), block(1024);
#define GMSM_SCATTER(D, CH)                                                                                              \
    hipLaunchKernelGGL((k_part_scatter<D, CH>), grid, block, scatter_lds, stream, (const D *)digits, n, nparts, fbits, lidx, \
                       pchunk_len, bh, part_base, parted)
            if (d16 && !big_chunk) GMSM_SCATTER(uint16_t, PART_CHUNK);
            else if (d16) GMSM_SCATTER(uint16_t, PART_CHUNK_BIG);
            else if (!big_chunk) GMSM_SCATTER(uint32_t, PART_CHUNK);
            else GMSM_SCATTER(uint32_t, PART_CHUNK_BIG);
#undef GMSM_SCATTER
        }
        // the partition's references stay in registers between the two passes: 24 per thread (stage_cap <= 24576), else 40
        if (stage_cap <= 24u * 1024u)
            hipLaunchKernelGGL(k_fine_sort<24>, dim3(nparts, nw), dim3(1024), ((size_t)4 << fbits) + (size_t)stage_cap * 4, stream,
                               parted, n, NB, fbits, lidx, part_base, sorted, starts, stage_cap);
        else
            hipLaunchKernelGGL(k_fine_sort<40>, dim3(nparts, nw), dim3(1024), ((size_t)4 << fbits) + (size_t)stage_cap * 4, stream,
                               parted, n, NB, fbits, lidx, part_base, sorted, starts, stage_cap);
        // partitions beyond the staging slots (crowded buckets, narrow top windows): sorted by many workgroups each. A run of at
        // most stage_cap entries per window cannot have one: no launches (14 us of a 0.4 ms call)
        if (n > stage_cap) {
            hipLaunchKernelGGL(k_heavy_hist, dim3(heavy_wg), dim3(1024), (size_t)4 << fbits, stream, parted, n, nw, nparts, fbits, lidx,
                               part_base, (const HeavyPart *)hparts, (const uint32_t *)hcount, hcap, scap, subhist);
            hipLaunchKernelGGL(k_heavy_scan, dim3(std::min<uint32_t>(hcap, 64u), nw), dim3(1024), (size_t)4 << fbits, stream, nparts, NB, fbits,
                               part_base, (const HeavyPart *)hparts, (const uint32_t *)hcount, hcap, scap, subhist, starts);
            hipLaunchKernelGGL(k_heavy_place, dim3(heavy_wg), dim3(1024), (size_t)4 << fbits, stream, parted, n, nw, nparts, fbits, lidx,
                               part_base, (const HeavyPart *)hparts, (const uint32_t *)hcount, hcap, scap, (const uint32_t *)subhist, sorted);
        }
        // ---- 2. bucket accumulation, 3. fix-up, 4. bucket reduction: for the windows [k0, k0 + nk) of the launch - all of
        // them, or (GMSM_OPT_SPLIT, experiment) two groups: the fix-up and reduction of the first group run on the merge stream
        // beside the accumulation of the second. Every array is window-major, so a group is the same kernels over shifted
        // base pointers; each group has its own long-chain counter and its own half of the list.
        if (forked) {
            HIP_TRY(hipStreamWaitEvent(stream, ws.ev_conv, 0));  // the rewritten bases are complete
            ws.conv_pending = false;
        }
        const auto accumulate = [&](uint32_t k0, uint32_t nk, hipStream_t st) {
            const uint32_t *st_g = starts + (size_t)k0 * (NB + 1);
            const uint32_t *sorted_g = sorted + (size_t)k0 * n;
            char *buckets_g = buckets + (size_t)k0 * NB * REC, *part_g = seg_partials + (size_t)k0 * qa.tpw * 2 * REC;
            if (shared)
                hipLaunchKernelGGL((k_accumulate_seg<U, true>), dim3((qa.tpw + 255) / 256, nk), dim3(256), 0, st, upoints, n, NB,
                                   qa.seg, st_g, sorted_g, buckets_g, part_g, seg_flags + (size_t)k0 * qa.tpw, seg_bucket + (size_t)k0 * qa.tpw,
                                   qa.tpw, (uint32_t)n_points, (uint32_t)resident->n);
            else
                hipLaunchKernelGGL((k_accumulate_seg<U, false>), dim3((qa.tpw + 255) / 256, nk), dim3(256), 0, st, upoints, n, NB,
                                   qa.seg, st_g, sorted_g, buckets_g, part_g, seg_flags + (size_t)k0 * qa.tpw, seg_bucket + (size_t)k0 * qa.tpw,
                                   qa.tpw, 0u, 0u);
        };
        const auto fixup_and_reduce = [&](uint32_t k0, uint32_t nk, size_t list_off, hipStream_t st, bool mark) {
            const uint32_t *st_g = starts + (size_t)k0 * (NB + 1);
            char *buckets_g = buckets + (size_t)k0 * NB * REC, *part_g = seg_partials + (size_t)k0 * qa.tpw * 2 * REC;
            const uint32_t *flags_g = seg_flags + (size_t)k0 * qa.tpw, *pb_g = seg_bucket + (size_t)k0 * qa.tpw;
            uint32_t *cnt_g = long_flag + k0;  // k_part_rowscan zeroed every window's slot
            LongChain *list_g = long_list + list_off;
            uint32_t *done_g = piece_done + list_off;
            void *sums_g = (char *)piece_sums + list_off * REC;
            if (shared && n / NB >= 2 * (size_t)qa.seg)  // buckets of several threads' worth of entries (dense chains): one thread per bucket
                hipLaunchKernelGGL((k_fixup_bucket<OpsSerial>), dim3((NB + 255) / 256, nk), dim3(256), 0, st, NB, st_g, qa.seg,
                                   (const void *)part_g, qa.tpw, (void *)buckets_g, cnt_g, list_g, done_g, FIX_MAXWALK);
            else if constexpr (FIXUP_QUAD)
                hipLaunchKernelGGL((k_fixup_seg_q<U>), dim3((qa.tpw + 63) / 64, nk), dim3(256), 128 * sizeof(QRec<U>), st, NB,
                                   (const void *)part_g, flags_g, pb_g, qa.tpw, (void *)buckets_g, cnt_g, list_g, done_g, FIX_MAXWALK, st_g, qa.seg);
            else
                hipLaunchKernelGGL((k_fixup_seg<OpsSerial>), dim3((qa.tpw + 255) / 256, nk), dim3(256), 0, st, NB, part_g, flags_g, pb_g,
                                   qa.tpw, buckets_g, cnt_g, list_g, done_g, FIX_MAXWALK, st_g, qa.seg);
            hipLaunchKernelGGL((k_fixup_long<U>), dim3(2 * ctx.num_cus), dim3(256), 128 * sizeof(QRec<U>), st, NB, part_g, pb_g, qa.tpw,
                               buckets_g, (const uint32_t *)cnt_g, (const LongChain *)list_g, done_g, sums_g, st_g, qa.seg);
            // ---- bucket reduction -> window totals (empty buckets are never written: the reduction consults starts[])
            if (mark) timer.mark(T_REDUCE, st);
            if (!buckets_only) enqueue_reduce_kernels(ctx, ws, q, T, buckets_g, st_g, nk, NB, st, k0);
        };
        timer.mark(T_ACCUMULATE, stream);
        if (!split) {
            accumulate(0, nw, stream);
            timer.mark(T_FIXUP, stream);
            fixup_and_reduce(0, nw, 0, stream, true);
        } else {
            const uint32_t ha = nw / 2;
            if ((rc = ws.side_stream(ws.mstream))) return rc;
            accumulate(0, ha, stream);
            HIP_TRY(hipEventRecord(ws.ev_buckets, stream));
            HIP_TRY(hipStreamWaitEvent(ws.mstream, ws.ev_buckets, 0));
            fixup_and_reduce(0, ha, 0, ws.mstream, false);
            HIP_TRY(hipEventRecord(ws.ev_merged, ws.mstream));
            accumulate(ha, nw - ha, stream);
            timer.mark(T_FIXUP, stream);
            fixup_and_reduce(ha, nw - ha, list_cap / 2, stream, true);
            HIP_TRY(hipStreamWaitEvent(stream, ws.ev_merged, 0));
        }
        if (!buckets_only && nwd > nw)
            hipLaunchKernelGGL((k_fill_infinity<Ext>), dim3((nwd - nw + 63) / 64), dim3(64), 0, stream, ws.totals.ptr, nw, nwd - nw);
        timer.mark(T_END, stream);
        HIP_TRY(hipGetLastError());
        if (!buckets_only) {
            if (d_out)
                HIP_TRY(hipMemcpyAsync(d_out, ws.totals.ptr, (size_t)nwd * sizeof(Ext), hipMemcpyDeviceToDevice, stream));
            else
                HIP_TRY(hipMemcpyAsync(ws.pinned, ws.totals.ptr, (size_t)nwd * sizeof(Ext), hipMemcpyDeviceToHost, stream));
        }
        if ((rc = end_use(ws, stream))) return rc;
        ws.pending_timed = timer.on;
        return GMSM_OK;
    }

    // The three kernels of the bucket reduction: buckets (nw x NB lazy records; starts == nullptr: every record is
    // stored, infinity as zz = 0) -> ws.totals. Scratch: ws.red_pre, ws.partials.
    static void enqueue_reduce_kernels(Context &ctx, Workspace &ws, const Geometry &q, uint32_t T, const void *buckets,
                                       const uint32_t *starts, uint32_t nw, uint32_t NB, hipStream_t stream,
                                       uint32_t k0 = 0 /* first window of a group: offsets into the scratch and the totals */) {
        constexpr size_t RECB = sizeof(typename OpsSerial::Mem);
        void *red_pre = (char *)ws.red_pre.ptr + (size_t)k0 * T * 2 * RECB;
        void *partials = (char *)ws.partials.ptr + (size_t)k0 * q.nblocks1 * 2 * RECB;  // (groups only with two levels)
        void *totals = (char *)ws.totals.ptr + (size_t)k0 * sizeof(Ext);
        // level 1 leaves S_blk already multiplied by the width of the level-2 spans (an otherwise idle quad doubles it
        // log2span times while the trees run): level 2 then has no serial doubling tail
        const uint32_t prescale = q.log2span;
        if constexpr (SERIAL_QUAD)
            hipLaunchKernelGGL((k_reduce_serial_q<U, SERIAL_Q_QUADS>), dim3((T + SERIAL_Q_QUADS - 1) / SERIAL_Q_QUADS, nw), dim3(4 * SERIAL_Q_QUADS),
                               3 * SERIAL_Q_QUADS * sizeof(QRec<U>), stream, buckets, NB, q.log2L, T, starts, red_pre);
        else
            hipLaunchKernelGGL((k_reduce_serial<OpsSerial>), dim3((T + 255) / 256, nw), dim3(256), 0, stream, buckets, NB,
                               q.log2L, T, starts, red_pre);
        // one combine level: `pairs` (S, W) pairs per window, each the sum of 2^l2 buckets (S prescaled by the caller's
        // factor when l2 == 0) -> `blocks` pairs per window
        const auto combine = [&](const void *in, uint32_t pairs, void *out, uint32_t blocks, uint32_t l2, uint32_t pre) {
            if constexpr (COMBINE_WE) {
                if (use_combine_we(ctx, blocks * nw)) {
                    hipLaunchKernelGGL((k_combine_we<U, true>), dim3(blocks, nw), dim3(4 * COMBINE_N),
                                       (2 * COMBINE_N + 1) * sizeof(QRec<U>), stream, l2, out, pre, in, pairs);
                    return;
                }
            }
            hipLaunchKernelGGL((k_combine_q<U, true, COMBINE_N>), dim3(blocks, nw), dim3(4 * COMBINE_N),
                               (2 * COMBINE_N + 1) * sizeof(QRec<U>), stream, l2, out, pre, in, pairs);
        };
        combine(red_pre, T, partials, q.nblocks1, q.log2L, prescale);
        const void *last = partials;
        uint32_t nlast = q.nblocks1, rest = q.log2span - prescale;
        if (q.nblocks2) {
            // second combine: its input pairs carry S already multiplied by their own span (the prescale above), so the
            // pairs count as single buckets (L = 1); its S_blk is doubled log2 N more times for the last level
            constexpr size_t REC = sizeof(typename OpsSerial::Mem);
            void *out2 = (char *)partials + (size_t)nw * q.nblocks1 * 2 * REC;
            uint32_t lgN = 0;
            for (size_t t = COMBINE_N; t > 1; t >>= 1) ++lgN;
            combine(partials, q.nblocks1, out2, q.nblocks2, 0, lgN);
            last = out2;
            nlast = q.nblocks2;
            rest = 0;
        }
        uint32_t active = 2;
        while (active < nlast) active <<= 1;
        hipLaunchKernelGGL((k_reduce2_q<U, true>), dim3(nw), dim3(4 * active), 2 * active * sizeof(QRec<U>), stream, last, nlast,
                           rest, active, totals);
        (void)ctx;
    }

    // The reduction alone, over bucket sums that a multi-range call has merged (every record stored): totals -> ws.pinned.
    static int enqueue_reduce(Context &ctx, Workspace &ws, const void *buckets, const WindowPlan &plan, size_t n_for_geometry,
                              hipStream_t stream) {
        const uint32_t nw = bucket_sets(plan), nwd = plan.nwin_local, NB = plan.nbuckets;
        constexpr size_t REC = sizeof(typename OpsSerial::Mem);
        const Geometry q = plan_geometry(ctx, nw, n_for_geometry, NB);
        const uint32_t T = (uint32_t)(((size_t)NB + ((size_t)1 << q.log2L) - 1) >> q.log2L);
        int rc;
        if ((rc = ws.partials.ensure((size_t)nw * (q.nblocks1 + q.nblocks2) * 2 * REC))) return rc;
        if ((rc = ws.red_pre.ensure((size_t)nw * T * 2 * REC))) return rc;
        if ((rc = ws.totals.ensure((size_t)nwd * sizeof(Ext)))) return rc;
        if ((rc = ws.ensure_pinned((size_t)nwd * sizeof(Ext)))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_combine_q<U, true, COMBINE_N>, (int)((2 * COMBINE_N + 1) * sizeof(QRec<U>))))) return rc;
        if constexpr (COMBINE_WE)
            if ((rc = ctx.allow_lds((const void *)k_combine_we<U, true>, (int)((2 * COMBINE_N + 1) * sizeof(QRec<U>))))) return rc;
        if constexpr (SERIAL_QUAD)
            if ((rc = ctx.allow_lds((const void *)k_reduce_serial_q<U, SERIAL_Q_QUADS>, (int)(3 * SERIAL_Q_QUADS * sizeof(QRec<U>))))) return rc;
        if ((rc = ctx.allow_lds((const void *)k_reduce2_q<U, true>, (int)(2 * RED2_TPB * sizeof(QRec<U>))))) return rc;
        enqueue_reduce_kernels(ctx, ws, q, T, buckets, nullptr, nw, NB, stream);
        if (nwd > nw)
            hipLaunchKernelGGL((k_fill_infinity<Ext>), dim3((nwd - nw + 63) / 64), dim3(64), 0, stream, ws.totals.ptr, nw, nwd - nw);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ws.pinned, ws.totals.ptr, (size_t)nwd * sizeof(Ext), hipMemcpyDeviceToHost, stream));
        return GMSM_OK;
    }

    // ---- N3: fixed-base batch scalar multiplication / batch normalisation (gmsm_fixedbase.h; the table is GroupHost's) ----
    // recs (lazy XYZZ records, n of them, already on the device in ws.buckets) -> d_out affine; K records per thread
    static int normalize_records(Workspace &ws, size_t n, void *d_out) {
        int rc;
        if ((rc = ws.partials.ensure(n * sizeof(U)))) return rc;
        constexpr int K = 32;
        const size_t threads = (n + K - 1) / K;
        hipLaunchKernelGGL((k_batch_normalize<U, INLINE_OPS, K>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ws.stream,
                           ws.buckets.ptr, n, (U *)ws.partials.ptr, d_out);
        HIP_TRY(hipGetLastError());
        return GMSM_OK;
    }

    // BatchScalarMultiplicationG1 (g1.go:1039): out[i] = scalars[i] * base, affine. d_scalars/d_out on the device.
    static int batch_scalar_mul(Context &ctx, Workspace &ws, const uint64_t *base_limbs, const void *d_scalars, size_t n,
                                void *d_out) {
        if (n == 0) return GMSM_OK;
        Aff base;
        memcpy(&base, base_limbs, sizeof base);
        if (base.is_infinity()) {
            HIP_TRY(hipMemsetAsync(d_out, 0, n * AFF_BYTES, ws.stream));
            return GMSM_OK;
        }
        // table size against additions per scalar: 2^7 x 32 windows up to 2^21 scalars, 2^10 x 24 beyond (BN254)
        const unsigned forced_c = options().fixed_base_bits.load(std::memory_order_relaxed);
        const unsigned c = forced_c ? forced_c : n < ((size_t)1 << 21) ? 8 : 11;
        const WindowPlan plan = make_plan(c, 0, 1);
        std::vector<Aff> table;
        build_fixed_base_table(base, plan.c, plan.nwin_total, plan.nbuckets, (int)std::min<unsigned>(16, usable_cpus()), table);
        int rc;
        const size_t tn = table.size();
        if ((rc = ws.h2d_points.ensure(tn * AFF_BYTES))) return rc;
        if ((rc = ws.upoints.ensure(tn * AFF_BYTES))) return rc;
        if ((rc = ws.skip.ensure(tn))) return rc;
        if ((rc = ws.buckets.ensure(n * sizeof(XYZZL<U>)))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, table.data(), tn * AFF_BYTES, hipMemcpyHostToDevice, ws.stream));
        hipLaunchKernelGGL((k_convert_points<U>), dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, ws.stream,
                           ws.h2d_points.ptr, tn, ws.upoints.ptr, (uint8_t *)ws.skip.ptr);
        hipLaunchKernelGGL((k_fixed_base<U, FrP, INLINE_OPS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ws.stream,
                           (const uint32_t *)d_scalars, n, plan, ws.upoints.ptr, ws.buckets.ptr);
        HIP_TRY(hipGetLastError());
        if ((rc = normalize_records(ws, n, d_out))) return rc;
        HIP_TRY(hipStreamSynchronize(ws.stream));  // `table` (pageable host memory) must outlive the copy
        (void)ctx;
        return GMSM_OK;
    }

    // BatchJacobianToAffineG1 (g1.go:989): d_jac = n Go-layout Jacobian points on the device -> d_out affine
    static int batch_jac_to_affine(Workspace &ws, const void *d_jac, size_t n, void *d_out) {
        if (n == 0) return GMSM_OK;
        int rc;
        if ((rc = ws.buckets.ensure(n * sizeof(XYZZL<U>)))) return rc;
        hipLaunchKernelGGL((k_jac_to_recs<U, INLINE_OPS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ws.stream, d_jac, n,
                           ws.buckets.ptr);
        HIP_TRY(hipGetLastError());
        return normalize_records(ws, n, d_out);
    }

    // Rewrites n Go-layout bases (device memory) into the lazy domain once; the result serves any number of MultiExp
    // calls over a prefix of the bases (kzg.Commit over pk.G1[:len(p)], ecc/bn254/kzg/kzg.go:159-176).
    static int register_bases(Context &ctx, const void *d_points, size_t n, hipStream_t stream, ResidentBases *out) {
        int rc;
        if ((rc = out->upoints.ensure(n * AFF_BYTES))) return rc;
        if ((rc = out->skip.ensure(n))) return rc;
        if (n) {
            hipLaunchKernelGGL((k_convert_points<U>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_points, n,
                               out->upoints.ptr, (uint8_t *)out->skip.ptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(stream));
        }
        out->n = n;
        (void)ctx;
        return GMSM_OK;
    }

    // ---- N4: point ingest (gmsm_ingest.h). Results of the checks come back through one 64-bit word:
    // (index << 3 | PointStatus) of the first offending point, all ones when every point passed.
    static int read_first_bad(Workspace &ws, long long *bad_index, uint32_t *status) {
        unsigned long long v = 0;
        HIP_TRY(hipMemcpyAsync(&v, ws.flagword.ptr, 8, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
        if (v == ~0ull) {
            *bad_index = -1;
            *status = PT_OK;
        } else {
            *bad_index = (long long)(v >> 3);
            *status = (uint32_t)(v & 7u);
        }
        return GMSM_OK;
    }
    // d_raw: n wire-format points on the device -> d_out: n Go-layout affine points (Montgomery)
    static int decode_raw(Workspace &ws, const void *d_raw, size_t n, int level, void *d_out, long long *bad_index,
                          uint32_t *status) {
        int rc;
        if ((rc = ws.flagword.ensure(8))) return rc;
        HIP_TRY(hipMemsetAsync(ws.flagword.ptr, 0xff, 8, ws.stream));
        if (n)
        {
            const dim3 grid((unsigned)((n + 127) / 128));
            if (level >= 3 && NEEDS_TORSION)
                hipLaunchKernelGGL((k_decode_raw<F, FrP, Consts, NEEDS_TORSION, NEEDS_TORSION>), grid, dim3(128), 0, ws.stream,
                                   (const uint8_t *)d_raw, n, level, (Aff *)d_out, (unsigned long long *)ws.flagword.ptr);
            else
                hipLaunchKernelGGL((k_decode_raw<F, FrP, Consts, NEEDS_TORSION, false>), grid, dim3(128), 0, ws.stream,
                                   (const uint8_t *)d_raw, n, level, (Aff *)d_out, (unsigned long long *)ws.flagword.ptr);
        }
        HIP_TRY(hipGetLastError());
        return read_first_bad(ws, bad_index, status);
    }
    // d_comp: n compressed points (Bytes(), sizeof(F) bytes each) -> d_out: n Go-layout affine points; level >= 2 adds the
    // subgroup check of the decoded points (the Decoder's default, marshal.go:300-330): a second launch, and the smaller
    // index of the two kinds of offender is the one reported (the reference stops at the first error of either kind).
    static int decode_compressed(Workspace &ws, const void *d_comp, size_t n, int level, void *d_out, long long *bad_index,
                                 uint32_t *status) {
        int rc;
        if ((rc = ws.flagword.ensure(8))) return rc;
        HIP_TRY(hipMemsetAsync(ws.flagword.ptr, 0xff, 8, ws.stream));
        if (n) {
            hipLaunchKernelGGL((k_decompress<F, Consts>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ws.stream,
                               (const uint8_t *)d_comp, n, (Aff *)d_out, (unsigned long long *)ws.flagword.ptr);
            // offenders were written as infinity, which passes: the word keeps the decoder's verdict for them
            if (level >= 2 && NEEDS_TORSION) launch_validate(ws, d_out, n, level);
        }
        HIP_TRY(hipGetLastError());
        return read_first_bad(ws, bad_index, status);
    }
    // (*G1Affine).Bytes over a vector: d_points -> d_comp (n * sizeof(F) bytes)
    static int encode_compressed(Workspace &ws, const void *d_points, size_t n, void *d_comp) {
        if (n)
            hipLaunchKernelGGL((k_compress<F, Consts>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ws.stream,
                               (const Aff *)d_points, n, (uint8_t *)d_comp);
        HIP_TRY(hipGetLastError());
        return GMSM_OK;
    }
    static void launch_validate(Workspace &ws, const void *d_points, size_t n, int level) {
        const dim3 grid((unsigned)((n + 127) / 128));
        if (level >= 3 && NEEDS_TORSION)  // the definition [r]P = infinity: its own kernel (gmsm_ingest.h, BY_DEF)
            hipLaunchKernelGGL((k_validate_points<F, FrP, Consts, NEEDS_TORSION, NEEDS_TORSION>), grid, dim3(128), 0, ws.stream,
                               (const Aff *)d_points, n, level, (unsigned long long *)ws.flagword.ptr);
        else
            hipLaunchKernelGGL((k_validate_points<F, FrP, Consts, NEEDS_TORSION, false>), grid, dim3(128), 0, ws.stream,
                               (const Aff *)d_points, n, level, (unsigned long long *)ws.flagword.ptr);
    }
    static int validate_points(Workspace &ws, const void *d_points, size_t n, int level, long long *bad_index,
                               uint32_t *status) {
        int rc;
        if ((rc = ws.flagword.ensure(8))) return rc;
        HIP_TRY(hipMemsetAsync(ws.flagword.ptr, 0xff, 8, ws.stream));
        if (n && level > 0) launch_validate(ws, d_points, n, level);
        HIP_TRY(hipGetLastError());
        return read_first_bad(ws, bad_index, status);
    }

    // Most points one pipeline run takes: the coarse partition pass keeps two 32-bit words per partition in LDS, which
    // caps it at 2^27 references per window. Larger inputs run as consecutive point ranges whose window totals are added
    // (the point decomposition of sharding.py, on one device). GMSM_OPT_MAX_RUN lowers the cap (tests).
    // A shared-bucket plan (window tables) sorts nwin * n entries as one window: a sort entry is 32 bits - the low
    // bucket bits of the coarse pass next to (index, sign) - and the coarse pass takes at most 2^13 partitions.
    static size_t max_run_points(const WindowPlan &plan = WindowPlan{0, 0, 0, 0, 1, 0, 0, 0}) {
        size_t cap = (size_t)1 << 27;
        if (plan.glv) cap >>= 1;  // two entries per point
        if (plan.shared) {
            uint32_t log2NB = 0;
            while ((1u << log2NB) < plan.nbuckets) ++log2NB;
            const uint32_t fb_min = log2NB > 13 ? log2NB - 13 : 0;
            cap = std::min(cap, (((size_t)1 << (31 - fb_min)) - 1) / plan.nwin_total);
        }
        const size_t forced = options().max_run.load(std::memory_order_relaxed);
        return forced ? std::min(cap, forced) : cap;
    }

    // ---- window tables of registered bases (gmsm_bases_precompute): table slab w = 2^(c w) P_i for every window w of the
    // c-bit decomposition, so that the digits of ALL windows index one bucket set - one reduction of 2^(c-1) buckets
    // instead of nwin of them, which in turn lets c grow (fewer windows = fewer additions). 288 GB of HBM pay for it:
    // nwin copies of the bases (BN254 G1, 2^20 points, c = 19: 14 x 64 MiB).
    static uint32_t bucket_sets(const WindowPlan &plan) { return plan.shared ? 1u : plan.nwin_local; }
    // GMSM_OPT_TABLES: 0 = never, 1 (default) = the measured range of call sizes, 2 = whenever the handle has tables (tests)
    static bool tables_serve(size_t n_registered, size_t n_call) {
        const unsigned mode = options().tables.load(std::memory_order_relaxed);
        if (mode == 0 || n_call == 0) return false;
        if (mode >= 2) return true;
        if (n_call < table_min_points() || n_call > table_max_points()) return false;
        return n_call * 16 >= n_registered;  // a short prefix of the bases: the tables' window is too wide for it
    }
    static bool use_tables(const ResidentBases *rb, size_t n) {
        if (!rb || rb->tab_c.load() == 0) return false;
        const unsigned forced = options().window_bits.load(std::memory_order_relaxed);
        if (forced >= 2 && forced <= 20 && forced != rb->tab_c.load()) return false;
        return tables_serve(rb->n, n);
    }
    // The plan of a MultiExp over n points: the measured window table, or the registered bases' tables when they exist
    static WindowPlan plan_for(const ResidentBases *rb, size_t n) {
        if (use_tables(rb, n)) {
            WindowPlan plan = make_plan(rb->tab_c.load(), 0, 1);
            plan.shared = 1;
            return plan;
        }
        if (rb == nullptr) {  // bases taken anew: GLV half scalars where they were measured ahead (glv_preferred_c)
            const unsigned mode = options().glv.load(std::memory_order_relaxed);
            const unsigned forced = options().window_bits.load(std::memory_order_relaxed);
            unsigned c = 0;
            if (mode >= 2) c = choose_c(FR_BITS, AFF_BYTES, 2 * n);           // always (A/B): the width of a 2 n-point call, or the forced one
            else if (mode == 1 && forced == 0) c = glv_preferred_c(FR_BITS, AFF_BYTES, n);
            if (c != 0 && glv_width_built(c)) return make_plan_glv(c, 0, 1);
        }
        return make_plan(choose_c(FR_BITS, AFF_BYTES, n), 0, 1);
    }
    // widths k_decompose_glv is instantiated for
    static bool glv_width_built(unsigned c) { return (c >= 10 && c <= 18) || c == 20; }
    // Default table width for n registered points, from sweeps of every group over 2^13..2^21 with the width forced
    // (tools/tables_sweep.py, profiles/r03_tables.log). What moves it away from the plain path's width: the shared set
    // holds nwin * n entries, so (a) a wider window pays twice - fewer slabs to add AND shorter chains of partial sums
    // per bucket (nwin n / 2^(c-1) entries per bucket; BW6-761 2^18: fix-up 1.83 ms at c = 14, 0.10 ms at c = 18) -,
    // (b) one reduction of 2^(c-1) buckets is cheap next to nwin of them, and (c) the TOP window must not be narrow: a
    // top window of t bits drops its n entries into 2^t buckets of the one set (BN254 at c = 19: 7 bits, a million
    // entries in 128 buckets, i.e. in one coarse partition - 0.9 ms of fine sort), which rules out 18, 19 for BN254,
    // 17, 18 for BLS12-381 (17 also fills the top window: the carry doubles the buckets) and 15..17 for BW6-761.
    static unsigned table_c(size_t n) {
        unsigned lg = 0;
        while (((size_t)2 << lg) <= n) ++lg;
        if (FR_BITS > 300) return 18;                                        // BW6-761
        if (AFF_BYTES == 64) return lg <= 14 ? 15 : lg <= 18 ? 16 : 17;      // BN254 G1
        if (FR_BITS == 254) return lg <= 16 ? 15 : lg <= 19 ? 16 : 20;       // BN254 G2
        return lg < 20 ? 16 : 20;                                            // BLS12-381 G1, G2
    }
    // Call sizes the tables serve (measured against the plain path, same sweeps): below, a call is all latency and the
    // plain path's narrow windows win; above, the one window of nwin * n entries costs more in the sort and in the chains
    // of partial sums than one reduction saves (BN254 G1 2^22: 6.02 against 5.90 ms), and wider tables would need sort
    // entries of more than 32 bits.
    static size_t table_min_points() { return (size_t)1 << (AFF_BYTES == 64 ? 13 : 15); }
    static size_t table_max_points() { return (size_t)3 << (FR_BITS == 255 && AFF_BYTES == 96 ? 19 : 20); }
    // slabs 1 .. nw-1 of a table over the first m registered bases (slab 0 = the bases themselves, copied): slab w = slab
    // w-1 doubled c times. `bad` != 0 afterwards: a multiple of a base reached the identity.
    static int build_table_slabs(Workspace &ws, const ResidentBases *rb, size_t m, unsigned c, uint32_t nw, void *tables, uint32_t *bad) {
        const size_t slab = m * AFF_BYTES;
        int rc;
        if ((rc = ws.buckets.ensure(m * sizeof(XYZZL<U>)))) return rc;
        if ((rc = ws.h2d_points.ensure(slab))) return rc;
        if ((rc = ws.skip.ensure(m))) return rc;
        if ((rc = ws.flagword.ensure(8))) return rc;
        HIP_TRY(hipMemsetAsync(ws.flagword.ptr, 0, 8, ws.stream));
        HIP_TRY(hipMemcpyAsync(tables, rb->upoints.ptr, slab, hipMemcpyDeviceToDevice, ws.stream));
        const dim3 grid((unsigned)((m + 255) / 256)), block(256);
        for (uint32_t w = 1; w < nw; ++w) {
            char *prev = (char *)tables + (size_t)(w - 1) * slab;
            hipLaunchKernelGGL((k_table_double<U, INLINE_OPS>), grid, block, 0, ws.stream, (const void *)prev,
                               (const uint8_t *)rb->skip.ptr, m, c, ws.buckets.ptr);
            if ((rc = normalize_records(ws, m, ws.h2d_points.ptr))) return rc;
            hipLaunchKernelGGL((k_convert_points<U>), grid, block, 0, ws.stream, (const void *)ws.h2d_points.ptr, m,
                               (void *)(prev + slab), (uint8_t *)ws.skip.ptr);
            hipLaunchKernelGGL(k_skip_mismatch, grid, block, 0, ws.stream, (const uint8_t *)rb->skip.ptr,
                               (const uint8_t *)ws.skip.ptr, m, (uint32_t *)ws.flagword.ptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(bad, ws.flagword.ptr, 4, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
        return GMSM_OK;
    }
    static constexpr unsigned SMALL_TABLE_C = 6;          // width of the narrow tables: 2^5 buckets, one lane quad each
    // ... over the first so many bases (43 x 4096 x 64 B = 11 MiB for BN254 G1). Measured against the other forms (profiles/r05_small_n.log,
    // last block): ahead up to 2^12 points for every group but BW6-761 (2^12: 2.36 ms against 2.08 for the sorted pipeline; 2^10: 0.70 against 1.07)
    static constexpr size_t SMALL_TABLE_POINTS = FR_BITS > 300 ? 2048 : 4096;
    static int precompute_tables(Context &ctx, Workspace &ws, ResidentBases *rb, unsigned c) {
        if (c == 0) c = table_c(rb->n);
        if (c < 2 || c > 20) return fail(GMSM_ERR_ARG, "table window width must be 2..20");
        if (rb->n == 0 || rb->n >= ((size_t)1 << 31)) return fail(GMSM_ERR_ARG, "window tables need 1 <= n < 2^31 bases");
        const WindowPlan plan = make_plan(c, 0, 1);
        const uint32_t nw = plan.nwin_total;
        const size_t n = rb->n, slab = n * AFF_BYTES;
        int rc;
        if ((rc = rb->tables.ensure((size_t)nw * slab))) return rc;
        if ((rc = order_after(ws, nullptr))) return rc;
        uint32_t bad = 0;
        if ((rc = build_table_slabs(ws, rb, n, c, nw, rb->tables.ptr, &bad))) return rc;
        if (bad) {  // a base of even order (not a subgroup point): its multiples reach the identity - no tables, plain path
            rb->tables.release();
            return fail(GMSM_ERR_ARG, "window tables: a multiple 2^k P of a base is the identity (bases outside the prime-order subgroup)");
        }
        rb->tab_nw = nw;
        rb->tab_c.store(c, std::memory_order_release);  // publication: everything above is complete (stream synchronised)
        // the narrow tables of the fused small-n kernel: calls of a few thousand points over these bases then fill ONE
        // bucket set per workgroup and need no host-side fold at all (enqueue_small, shared form)
        {
            const WindowPlan sp = make_plan(SMALL_TABLE_C, 0, 1);
            const size_t m = std::min<size_t>(n, SMALL_TABLE_POINTS);
            if (rb->small_tables.ensure((size_t)sp.nwin_total * m * AFF_BYTES) == GMSM_OK &&
                build_table_slabs(ws, rb, m, SMALL_TABLE_C, sp.nwin_total, rb->small_tables.ptr, &bad) == GMSM_OK && !bad) {
                rb->small_nw = sp.nwin_total;
                rb->small_m = m;
                rb->small_c.store(SMALL_TABLE_C, std::memory_order_release);
            } else {
                // the narrow tables are an extra: without them small calls over the handle take the plain fused kernel. Their
                // memory goes back and the failure's text does not linger as this thread's last error (the call succeeded).
                rb->small_tables.release();
                clear_last_error();
            }
        }
        (void)ctx;
        return GMSM_OK;
    }

    // Point ranges of a MultiExp over device-resident inputs. More than one range (a) beyond the cap of one pipeline run
    // and (b) - experiment, GMSM_DEVICE_RANGES - to keep the bases of a range inside the 256 MiB Infinity Cache while
    // all windows gather from them.
    static unsigned device_ranges(size_t n, const WindowPlan &plan) {
        size_t run = max_run_points(plan);
        // BN254 G1 beyond 2^24 points: ranges of 2^24. The coarse sort of a longer run has 2048+ partitions and writes
        // 8-entry runs (scatter + fine sort 4.2 ms at 2^25, 9.0 at 2^26, against 1.45 per 2^24 points); the ranges share one
        // bucket set and one reduction, a merge costs 0.1 ms: 2^25 43.8 -> 42.0 ms, 2^26 86.3 -> 83.0 (profiles/r05_device_ranges.log).
        // Neutral for BLS12-381 G1 (84.6 / 84.4-84.7 ms at 2^25) and worse below 2^24 (BN254 G1 2^24 in two ranges: 21.5 -> 22.0).
        if (AFF_BYTES == 64 && !plan.shared && options().max_run.load(std::memory_order_relaxed) == 0) run = std::min<size_t>(run, (size_t)1 << 24);
        unsigned nr = (unsigned)((n + run - 1) / run);
        const unsigned forced = GMSM_TUNE(DEVICE_RANGES, 0);
        if (forced > nr) nr = (unsigned)std::min<size_t>(forced, n);
        return nr;
    }

    // ---- the fused small-n kernel (gmsm_small.h): calls of at most small_max_points() points that neither force a window
    // width nor are served by window tables. The width minimises the depth of the kernel's dependency chain -
    // log2(points per bucket) + 2 (c - 1) additions - against the number of windows the host has to fold.
    static constexpr uint32_t SMALL_SL = SmallSlice<U>::value;
    static constexpr bool SMALL_QUAD_ONLY = SmallQuadOnly<U>::value;
    // Measured against the sorted pipeline (profiles/r05_small_n.log, resident ms, fused / pipeline): BN254 G1 2^5 0.147 / 0.27,
    // 2^10 0.18 / 0.29-0.35, 2^12 0.28 / 0.41, 2^13 0.38 / 0.47; BN254 G2 2^11 0.65 / 0.91, 2^13 1.29 / 0.99; BLS12-381 G1 2^11 0.41 /
    // 0.62, 2^12 0.50 / 0.65, 2^13 0.73 / 0.70; BLS12-381 G2 2^11 1.19 / 1.62, 2^12 1.68 / 1.65; BN254 G2 2^12 0.86 / 0.86; BW6-761 2^11 1.57 / 2.05,
    // 2^12 2.5 / 2.1. Round 6 (GLV half scalars, lane quads for the bucket phase): profiles/r06_small_n.log.
    // Largest call the fused kernel takes, from the same-process sweeps of profiles/r06_small_forms.log (fused / sorted pipeline,
    // resident ms): BN254 G1 2^13 0.36 / 0.46; BLS12-381 G1 2^12 0.47 / 0.65, 2^13 0.69 / 0.69; BN254 G2 2^12 0.83 / 0.88, 2^13
    // 1.12 / 0.90; BLS12-381 G2 2^10 0.98 / 1.87, 2^11 1.60-1.68 / 1.63; BW6-761 2^10 1.61 / 2.41, 2^11 2.73 / 2.07 (the quad form
    // keeps one workgroup per CU at 310 registers: beyond ~1000 points the wide types are throughput-bound there).
    static size_t small_max_points() {
        const size_t forced = options().small_max.load(std::memory_order_relaxed);
        return std::min<size_t>(forced ? forced : GMSM_TUNE(SMALL_MAX, sizeof(U) <= 36 ? 8192 : sizeof(U) <= 72 ? 4096 : 1024),
                                small_entry_cap() / 2);
    }
    // entries one window's slices can hold: 64 slices of 256 entries (one-lane form) or of 8 chunks of 64 (quad form)
    static constexpr size_t small_entry_cap() {
        return SMALL_QUAD_ONLY ? (size_t)SMALL_QUAD_ENTRIES * SMALL_QUAD_MAX_CHUNKS * SMALL_MAX_SLICES : (size_t)SMALL_SL * SMALL_MAX_SLICES;
    }
    // GLV half scalars in the fused kernel: unless GMSM_OPT_GLV = 0
    static bool small_glv() { return options().glv.load(std::memory_order_relaxed) >= 1; }
    static WindowPlan small_make_plan(unsigned c, bool glv) { return glv ? make_plan_glv(c, 0, 1) : make_plan(c, 0, 1); }
    static unsigned small_c(size_t n, bool glv) {
        const unsigned forced = options().small_bits.load(std::memory_order_relaxed);
        // flat over 5..7 for the narrow types (fewer windows = less host fold, more buckets = more quad steps); the 28-limb
        // field and large calls take 7
        const size_t entries = glv ? 2 * n : n;
        unsigned c = (forced >= 2 && forced <= SMALL_MAX_C) ? forced
                     : FR_BITS > 300 ? (entries <= 512 ? 6 : 7)
                                     : entries <= 256 ? 5 : entries <= 4096 ? 6 : 7;
        // the kernel holds one lane quad per bucket: a width whose top window needs c + 1 bits (the scalar's bit
        // length a multiple of c) doubles the buckets - step down until they fit
        while (c > 2 && small_make_plan(c, glv).nbuckets > SMALL_NB_MAX) --c;
        return c;
    }
    static bool small_serves(size_t n, const ResidentBases *rb) {
        if (options().small_bits.load(std::memory_order_relaxed) == 1) return false;
        if (options().window_bits.load(std::memory_order_relaxed) != 0) return false;  // a forced width: the sorted pipeline
        if (small_shared(n, rb)) return true;
        return n >= 1 && n <= small_max_points() && !use_tables(rb, n);
    }
    // the narrow window tables of registered bases serve the call: every (window, point) pair is an entry of ONE bucket set
    static bool small_shared(size_t n, const ResidentBases *rb) {
        if (!rb || n < 1 || options().tables.load(std::memory_order_relaxed) == 0) return false;
        const unsigned c = rb->small_c.load(std::memory_order_acquire);
        if (c == 0 || n > rb->small_m) return false;
        const unsigned forced = options().small_bits.load(std::memory_order_relaxed);
        if (forced >= 2 && forced != c) return false;
        return n * rb->small_nw <= (size_t)SMALL_SL * SMALL_SHARED_MAX_SLICES;
    }
    // What one call of the fused kernel looks like; decided ONCE per call (options and the publication of the narrow tables
    // are read here and nowhere else on the call's path).
    struct SmallPlan {
        WindowPlan plan;
        bool shared, glv, quad;
        uint32_t chunks, nslices;
    };
    static SmallPlan small_plan(size_t n, const ResidentBases *rb) {
        SmallPlan sp;
        sp.shared = small_shared(n, rb);
        sp.glv = !sp.shared && small_glv();
        sp.plan = sp.shared ? make_plan(rb->small_c.load(), 0, 1) : small_make_plan(small_c(n, sp.glv), sp.glv);
        const size_t entries = sp.shared ? n * sp.plan.nwin_total : (sp.glv ? 2 * n : n);
        // the bucket phase on lane quads (64 entries per chunk): always for the wide element types; for the narrow ones while
        // the call is so small that a workgroup per 64 entries still leaves CUs idle (GMSM_OPT_SMALL_QUAD: 0 = this rule,
        // 1 = never, 2 = always)
        const unsigned fq = options().small_quad.load(std::memory_order_relaxed);
        const size_t wgs64 = (entries + SMALL_QUAD_ENTRIES - 1) / SMALL_QUAD_ENTRIES * (sp.shared ? 1 : sp.plan.nwin_total);
        sp.quad = SMALL_QUAD_ONLY || fq == 2 || (fq == 0 && wgs64 <= (size_t)GMSM_TUNE(SMALL_QUAD_WGS, 512));
        if (sp.quad) {
            const size_t max_slices = sp.shared ? SMALL_SHARED_MAX_SLICES : SMALL_MAX_SLICES;
            size_t chunks = 1;
            while ((entries + chunks * SMALL_QUAD_ENTRIES - 1) / (chunks * SMALL_QUAD_ENTRIES) > max_slices) chunks *= 2;
            // every workgroup ends with the same 2 (c - 1) + log2(slices) reduction steps whatever it accumulated: once the
            // launch has more workgroups than the chip holds at a time (one per CU for the wide types: 310 registers, 89 KB of
            // LDS), longer walks per workgroup beat more workgroups (BW6-761 2^10: 896 workgroups of one chunk 1.61 ms)
            const size_t nsets = sp.shared ? 1 : sp.plan.nwin_total, resident_wgs = SMALL_QUAD_ONLY ? 256 : 768;
            while (chunks < SMALL_QUAD_MAX_CHUNKS &&
                   (entries + chunks * SMALL_QUAD_ENTRIES - 1) / (chunks * SMALL_QUAD_ENTRIES) * nsets > resident_wgs &&
                   entries > chunks * SMALL_QUAD_ENTRIES)
                chunks *= 2;
            sp.chunks = (uint32_t)chunks;
            sp.nslices = (uint32_t)((entries + chunks * SMALL_QUAD_ENTRIES - 1) / (chunks * SMALL_QUAD_ENTRIES));
        } else {
            sp.chunks = 0;
            sp.nslices = (uint32_t)((entries + SMALL_SL - 1) / SMALL_SL);
        }
        return sp;
    }
    // Enqueues the kernel on ws.stream; the window totals (shared form: ONE total) land in ws.pinned. d_points == nullptr:
    // `resident`.
    static int enqueue_small(Context &ctx, Workspace &ws, const void *d_points, const void *d_scalars, size_t n,
                             const SmallPlan &sp, const ResidentBases *resident) {
        const WindowPlan &plan = sp.plan;
        const uint32_t nw = sp.shared ? 1u : plan.nwin_total;
        constexpr size_t REC = sizeof(typename OpsSerial::Mem);
        int rc;
        ws.pending_timed = false;
        if ((rc = begin_use(ws, ws.stream))) return rc;
        if ((rc = ws.ensure_pinned((size_t)plan.nwin_total * sizeof(Ext)))) return rc;
        if ((rc = ws.small_sums.ensure((size_t)nw * sp.nslices * REC))) return rc;
        if ((rc = ws.small_done.ensure((size_t)HEAVY_MAX_WINDOWS * 4))) return rc;
        // the slice counters start at zero: cleared before every launch that uses them (a launch that died half-way must not
        // leave a later call without its last workgroup - advisor, round 5)
        if (sp.nslices > 1) HIP_TRY(hipMemsetAsync(ws.small_done.ptr, 0, (size_t)nw * 4, ws.stream));
        SmallArgs a;
        a.points = d_points;
        a.upoints = nullptr;
        a.skip = nullptr;
        if (d_points == nullptr) {
            a.upoints = sp.shared ? resident->small_tables.ptr : resident->upoints.ptr;
            a.skip = (const uint8_t *)resident->skip.ptr;
        }
        a.scalars = (const uint32_t *)d_scalars;
        a.n = (uint32_t)n;
        a.glv = sp.glv ? 1u : 0u;
        a.tab_m = sp.shared ? (uint32_t)resident->small_m : 0u;
        a.chunks = sp.chunks;
        a.slice_sums = ws.small_sums.ptr;
        a.done = (uint32_t *)ws.small_done.ptr;
        // the window totals go straight into the pinned result buffer (host memory mapped into the device: nwin 128-byte
        // stores over PCIe instead of a copy kernel and its launch, 5-8 us of a 0.15 ms call)
        a.totals = ws.pinned;
        g_small_runs.fetch_add(1, std::memory_order_relaxed);
        if (sp.shared) g_table_runs.fetch_add(1, std::memory_order_relaxed);
        // profiling: the one launch counts as the call's accumulation stage (two events, whatever the level)
        StageTimer timer(ws, true);
        ws.timed = timer.on;
        if (timer.on) {
            ws.timed_level = 2;
            (void)hipEventRecord(ws.events[T_ACCUMULATE], ws.stream);
        }
        const dim3 grid(sp.nslices, nw);
        if (sp.quad) {
            const size_t lds = (size_t)192 * sizeof(QRec<U>);
            if (sp.shared) {
                if ((rc = ctx.allow_lds((const void *)k_msm_small_q<U, FrP, Consts, true>, (int)lds))) return rc;
                hipLaunchKernelGGL((k_msm_small_q<U, FrP, Consts, true>), grid, dim3(256), lds, ws.stream, a, plan);
            } else {
                if ((rc = ctx.allow_lds((const void *)k_msm_small_q<U, FrP, Consts, false>, (int)lds))) return rc;
                hipLaunchKernelGGL((k_msm_small_q<U, FrP, Consts, false>), grid, dim3(256), lds, ws.stream, a, plan);
            }
        } else {
            if constexpr (!SMALL_QUAD_ONLY) {
                const size_t lds = (size_t)SMALL_SL * REC;
                if (sp.shared) {
                    if ((rc = ctx.allow_lds((const void *)k_msm_small<U, FrP, Consts, SMALL_SL, true>, (int)lds))) return rc;
                    hipLaunchKernelGGL((k_msm_small<U, FrP, Consts, SMALL_SL, true>), grid, dim3(SMALL_SL), lds, ws.stream, a, plan);
                } else {
                    if ((rc = ctx.allow_lds((const void *)k_msm_small<U, FrP, Consts, SMALL_SL, false>, (int)lds))) return rc;
                    hipLaunchKernelGGL((k_msm_small<U, FrP, Consts, SMALL_SL, false>), grid, dim3(SMALL_SL), lds, ws.stream, a, plan);
                }
            }
        }
        HIP_TRY(hipGetLastError());
        if (timer.on) (void)hipEventRecord(ws.events[T_FIXUP], ws.stream);
        ws.pending_timed = timer.on;
        return end_use(ws, ws.stream);
    }
    // waits for an enqueue_small and turns what it left in ws.pinned into the result
    static int collect_small(Workspace &ws, const SmallPlan &sp, J *out) {
        if (sp.shared) {  // one total, nothing to fold
            Ext total;
            int rc = collect_window_sums(ws, ws.stream, 1, &total);
            if (rc) return rc;
            *out = total.zz.is_zero() ? J{F::one(), F::one(), F::zero()} : jac_from_xyzz(total);
            return GMSM_OK;
        }
        std::vector<Ext> totals(sp.plan.nwin_total);
        int rc = collect_window_sums(ws, ws.stream, sp.plan.nwin_total, totals.data());
        if (rc) return rc;
        *out = fold(totals.data(), sp.plan.c, sp.plan.nwin_total);
        return GMSM_OK;
    }
    static int multiexp_small(Context &ctx, Workspace &ws, const void *d_points, const void *d_scalars, size_t n,
                              hipStream_t caller_stream, J *out, const ResidentBases *resident) {
        const SmallPlan sp = small_plan(n, d_points ? nullptr : resident);
        if (sp.plan.nwin_total > HEAVY_MAX_WINDOWS) return fail(GMSM_ERR_ARG, "small path: too many windows");
        int rc = order_after(ws, caller_stream);
        if (rc) return rc;
        if ((rc = enqueue_small(ctx, ws, d_points, d_scalars, n, sp, resident))) return rc;
        return collect_small(ws, sp, out);
    }

    static int multiexp_device(Context &ctx, Workspace &ws, const void *d_points, const void *d_scalars, size_t n,
                               hipStream_t caller_stream, J *out, const ResidentBases *resident = nullptr) {
        if (small_serves(n, resident)) return multiexp_small(ctx, ws, d_points, d_scalars, n, caller_stream, out, resident);
        const WindowPlan plan = plan_for(resident, n);
        const unsigned c = plan.c;
        const unsigned nr = device_ranges(n, plan);
        std::vector<Ext> totals(plan.nwin_total);
        if (nr <= 1) {
            int rc = window_sums(ctx, ws, d_points, d_scalars, n, plan, caller_stream, totals.data(), resident);
            if (rc) return rc;
            *out = fold(totals.data(), c, plan.nwin_total);
            return GMSM_OK;
        }
        // Consecutive point ranges on the workspace's stream: each leaves its bucket sums, k_merge_buckets adds them to the
        // running buckets, one reduction at the end (the reference's split + AddAssign, multiexp.go:98-140, bucket by bucket).
        constexpr size_t REC = sizeof(typename OpsSerial::Mem);
        const size_t per = (n + nr - 1) / nr;
        int rc = order_after(ws, caller_stream);
        if (rc) return rc;
        if ((rc = ws.carry.ensure((size_t)bucket_sets(plan) * plan.nbuckets * REC))) return rc;
        // stage profile of a multi-range call: every range's events are collected before the next range reuses them (one
        // stream synchronisation per range, profiling only), the call counts once; the final reduction is not in it
        const bool prof = profiling_level() != 0;
        float pms[STAGE_COUNT] = {0};
        unsigned plaunch[STAGE_COUNT] = {0};
        for (unsigned r = 0; r * per < n; ++r) {
            const size_t lo = (size_t)r * per, len = std::min(per, n - lo);
            const void *dp = d_points ? (const char *)d_points + lo * AFF_BYTES : nullptr;
            if ((rc = enqueue_window_sums(ctx, ws, dp, (const char *)d_scalars + lo * SCALAR_BYTES, len, plan, ws.stream, resident,
                                          nullptr, lo, /*buckets_only=*/true, /*time_range=*/prof)))
                return rc;
            if (prof && ws.pending_timed) {
                HIP_TRY(wait_stream(ws.stream));
                StageTimer::collect_add(ws, pms, plaunch);
                ws.pending_timed = false;
            }
            hipLaunchKernelGGL((k_merge_buckets<OpsSerial>), dim3((plan.nbuckets + 255) / 256, bucket_sets(plan)), dim3(256), 0, ws.stream,
                               ws.carry.ptr, (const void *)ws.buckets.ptr, (const uint32_t *)ws.starts.ptr, plan.nbuckets, r == 0 ? 1 : 0);
        }
        if ((rc = enqueue_reduce(ctx, ws, ws.carry.ptr, plan, per, ws.stream))) return rc;
        if ((rc = collect_window_sums(ws, ws.stream, plan.nwin_local, totals.data()))) return rc;
        if (prof) record_stage_times(pms, plaunch);
        *out = fold(totals.data(), c, plan.nwin_total);
        return GMSM_OK;
    }

    // Asynchronous pair (gmsm_multiexp_bases_submit / gmsm_multiexp_collect): submit launches the whole device pipeline
    // on the workspace's own stream and returns; collect waits for it, then folds the windows on the host.
    static int multiexp_submit(Context &ctx, Workspace &ws, const void *d_scalars, size_t n, const ResidentBases *resident) {
        if (n > max_run_points())
            return fail(GMSM_ERR_ARG, "submit/collect takes at most 2^27 points per ticket: use the blocking entry, which splits larger inputs");
        WindowPlan plan = plan_for(resident, n);
        if (n > max_run_points(plan)) plan = make_plan(choose_c(FR_BITS, AFF_BYTES, n), 0, 1);  // one run per ticket: plain path
        const unsigned c = plan.c;
        int rc = enqueue_window_sums(ctx, ws, nullptr, d_scalars, n, plan, ws.stream, resident);
        if (rc) return rc;
        ws.pending_c = c;
        ws.pending_nw = plan.nwin_total;
        return GMSM_OK;
    }
    static int multiexp_collect(Workspace &ws, J *out) {
        std::vector<Ext> totals(ws.pending_nw);
        int rc = collect_window_sums(ws, ws.stream, ws.pending_nw, totals.data());
        if (rc) return rc;
        *out = fold(totals.data(), ws.pending_c);
        return GMSM_OK;
    }


    // Number of point ranges a host-buffer MultiExp is cut into so that the H2D copy of range k+1 runs under the
    // pipeline of range k (two workspaces, two streams). Round 2 reduced every range on its own and added the totals on
    // the host: a range then paid the size-independent part of the pipeline again (0.35 ms reduction chain), so 2^20 ran
    // as 2 ranges (cold 3.99 / 3.31 / 3.94 ms with 1 / 2 / 4 ranges, profiles/r02_host_ranges.log). Now the ranges share one
    // bucket set and ONE reduction (k_merge_buckets), a range costs its share of the accumulation plus ~0.1 ms, and the
    // call is cut finer: what is exposed is the copy of the first range and the tail after the last.
    // GMSM_OPT_HOST_RANGES overrides.
    static unsigned host_ranges(size_t n, bool with_points) {
        const unsigned forced = options().host_ranges.load(std::memory_order_relaxed);
        if (forced) return (unsigned)std::min<size_t>(forced, std::max<size_t>(1, n));
        // Measured (profiles/r03_host_ranges.log, BN254 G1, cold / warm-bases ms): 2^20 1 range 3.73 / 2.51, 2: 3.19 / 2.23,
        // 4: 2.95 / 2.31, 8: 3.38 / 2.70; 2^22 4: 9.47 / 6.97, 8: 8.87 / 7.11, 16: 9.69 / 8.26; 2^24 8: 32.5 / 23.5, 16: 31.5 / 23.3,
        // 32: 31.8 / 25.1 (resident 2.00 / 6.25 / 22.1): a range costs about 0.1 ms of launches and merge.
        if (with_points) {  // bases + scalars cross PCIe (96 B per BN254 G1 point)
            if (n < ((size_t)1 << 19)) return 1;
            if (n < ((size_t)1 << 23)) return (unsigned)std::min<size_t>(8, n >> 18);
            return 16;
        }
        if (n < ((size_t)1 << 20)) return 1;  // scalars only (32 B per point)
        // the two G1 groups of 64 / 96-byte points: three to eight ranges, the first (and, from four, the last) half as long
        // (window_sums_from_host); the wider groups compute 3-10 times longer per point - the copy hardly shows, every
        // range costs its launches: few, uniform ranges (BN254 G2 2^20: 6.19 ms with two, 6.43 with three skewed)
        if (SKEWED_HOST_RANGES) return (unsigned)std::min<size_t>(8, std::max<size_t>(3, n >> 21));
        return (unsigned)std::min<size_t>(16, std::max<size_t>(2, n >> 20));
    }

    // Window totals of a MultiExp whose scalars (and, unless `resident`, points) are in host memory: the windows of `plan`
    // over points [0, n) (registered bases: [resident_base, resident_base + n)), cut into point ranges whose copies run
    // under the pipeline of the previous range; out_totals receives plan.nwin_local totals (the ranges' totals added on the
    // host, g1JacExtended.add). `first` is leased by the caller; a second workspace is borrowed when one is free,
    // otherwise the ranges run one after the other.
    static int window_sums_from_host(Context &ctx, Workspace &first, const uint64_t *points, const ResidentBases *resident,
                                     size_t resident_base, const uint64_t *scalars, size_t n, const WindowPlan &plan,
                                     unsigned nr, Ext *out_totals) {
        const uint32_t nw = plan.nwin_local, nsets = bucket_sets(plan);
        if (nw == 0) return GMSM_OK;
        const size_t per = (n + nr - 1) / nr;
        // Range boundaries. With registered bases only the scalars cross PCIe and the device is the slower side: what the
        // call exposes of its copies is the copy of the FIRST range (nothing to compute yet), so that one is half as long
        // as the others - and so is the last one once there are many (its pipeline is the tail after the last copy).
        // Measured (profiles/r03_host_skew.log, BN254 G1 warm-bases, uniform -> skewed): 2^20 2.27 -> 2.15 ms, 2^22 6.92 -> 6.67,
        // 2^24 23.4 -> 22.5; with the bases crossing too (cold) the link is as slow as the device and uniform ranges are as
        // good as any. A forced count (GMSM_OPT_HOST_RANGES) keeps uniform ranges.
        std::vector<size_t> cut(nr + 1, 0);
        {
            const bool skew = SKEWED_HOST_RANGES && points == nullptr && nr >= 3 && options().host_ranges.load(std::memory_order_relaxed) == 0;
            std::vector<double> wgt(nr, 1.0);
            if (skew) {
                wgt[0] = 0.5;
                if (nr >= 4) wgt[nr - 1] = 0.5;
            }
            double tot = 0, run = 0;
            for (double v : wgt) tot += v;
            bool ok = true;
            for (unsigned r = 0; r < nr; ++r) {
                run += wgt[r];
                cut[r + 1] = r + 1 == nr ? n : (size_t)((double)n * run / tot);
                if (cut[r + 1] <= cut[r] || cut[r + 1] - cut[r] > max_run_points(plan)) ok = false;
            }
            if (!ok)
                for (unsigned r = 0; r <= nr; ++r) cut[r] = std::min(n, (size_t)r * per);
        }
        Workspace *w[2] = {&first, nr > 1 ? ctx.acquire(false) : nullptr};
        const unsigned nws = w[1] ? 2 : 1;
        // More than one range: every range stops after its fix-up, its bucket sums are added to the call's running buckets
        // (first.carry) on a third stream, and ONE reduction follows the last merge - a range costs its accumulation and
        // one addition per occupied bucket, not another 0.35 ms reduction chain. With one workspace the ranges and the
        // merges simply alternate on its stream.
        constexpr size_t REC = sizeof(typename OpsSerial::Mem);
        int rc = GMSM_OK;
        if (nws == 2 && (rc = first.side_stream(first.mstream))) {
            ctx.release(w[1]);
            return rc;
        }
        hipStream_t ms = nws == 2 ? first.mstream : first.stream;
        if (nr > 1 && (rc = first.carry.ensure((size_t)nsets * plan.nbuckets * REC))) {
            if (w[1]) ctx.release(w[1]);
            return rc;
        }
        for (unsigned r = 0; r < nr && rc == GMSM_OK; ++r) {
            Workspace &ws = *w[r % nws];
            const size_t lo = cut[r], len = cut[r + 1] - lo;
            const void *dp = nullptr;
            // hipMemcpyAsync from pageable memory returns when the caller's buffer has been consumed; the kernels
            // queued behind it do not wait for the host, so range k computes while range k+1 is being copied
            if ((rc = ws.h2d_scalars.ensure(len * SCALAR_BYTES))) break;
            if (hipMemcpyAsync(ws.h2d_scalars.ptr, (const char *)scalars + lo * SCALAR_BYTES, len * SCALAR_BYTES,
                               hipMemcpyHostToDevice, ws.stream) != hipSuccess) {
                rc = fail(GMSM_ERR_DEVICE, "hipMemcpyAsync(scalars) failed");
                break;
            }
            if (points) {
                if ((rc = ws.h2d_points.ensure(len * AFF_BYTES))) break;
                if (hipMemcpyAsync(ws.h2d_points.ptr, (const char *)points + lo * AFF_BYTES, len * AFF_BYTES,
                                   hipMemcpyHostToDevice, ws.stream) != hipSuccess) {
                    rc = fail(GMSM_ERR_DEVICE, "hipMemcpyAsync(points) failed");
                    break;
                }
                dp = ws.h2d_points.ptr;
            }
            if (nr == 1) {
                rc = enqueue_window_sums(ctx, ws, dp, ws.h2d_scalars.ptr, len, plan, ws.stream, resident, nullptr, resident_base + lo);
                break;
            }
            // the buckets of the range this workspace ran two ranges ago must have been merged before they are overwritten
            hipError_t he = hipSuccess;
            if (nws == 2 && r >= 2) he = hipStreamWaitEvent(ws.stream, ws.ev_merged, 0);
            if (he != hipSuccess) {
                rc = fail(GMSM_ERR_DEVICE, std::string("hipStreamWaitEvent: ") + hipGetErrorString(he));
                break;
            }
            if ((rc = enqueue_window_sums(ctx, ws, dp, ws.h2d_scalars.ptr, len, plan, ws.stream, resident, nullptr,
                                          resident_base + lo, /*buckets_only=*/true)))
                break;
            if (nws == 2) {
                he = hipEventRecord(ws.ev_buckets, ws.stream);
                if (he == hipSuccess) he = hipStreamWaitEvent(ms, ws.ev_buckets, 0);
            }
            hipLaunchKernelGGL((k_merge_buckets<OpsSerial>), dim3((plan.nbuckets + 255) / 256, nsets), dim3(256), 0, ms, first.carry.ptr,
                               (const void *)ws.buckets.ptr, (const uint32_t *)ws.starts.ptr, plan.nbuckets, r == 0 ? 1 : 0);
            if (he == hipSuccess && nws == 2) he = hipEventRecord(ws.ev_merged, ms);
            if (he == hipSuccess) he = hipGetLastError();
            if (he != hipSuccess) rc = fail(GMSM_ERR_DEVICE, std::string("multi-range merge: ") + hipGetErrorString(he));
        }
        if (rc == GMSM_OK && nr > 1) rc = enqueue_reduce(ctx, first, first.carry.ptr, plan, per, ms);
        if (rc == GMSM_OK && wait_stream(ms) != hipSuccess) rc = fail(GMSM_ERR_DEVICE, "hipStreamSynchronize failed");
        for (unsigned i = 0; i < nws; ++i) {  // nothing of this call is left running on either workspace
            (void)hipStreamSynchronize(w[i]->stream);
            if (rc == GMSM_OK && w[i]->pending_timed) StageTimer::collect(*w[i]);
            w[i]->pending_timed = false;
        }
        if (w[1]) ctx.release(w[1]);
        if (rc) return rc;
        memcpy(out_totals, first.pinned, (size_t)nw * sizeof(Ext));
        return GMSM_OK;
    }

    // Point ranges of a host-buffer call over n points: host_ranges(), more if one of them would exceed a pipeline run.
    static unsigned host_range_count(size_t n, bool with_points, const WindowPlan &plan) {
        unsigned nr = host_ranges(n, with_points);
        const size_t run = max_run_points(plan);
        if ((n + nr - 1) / nr > run) nr = (unsigned)((n + run - 1) / run);
        const size_t per = (n + nr - 1) / nr;
        return (unsigned)((n + per - 1) / per);
    }

    // MultiExp with the scalars (and, unless `resident`, the points) in host memory.
    static int multiexp_from_host(Context &ctx, Workspace &first, const uint64_t *points, const ResidentBases *resident,
                                  const uint64_t *scalars, size_t n, J *out) {
        if (small_serves(n, resident)) {  // one copy, one launch (gmsm_small.h)
            Workspace &ws = first;
            int rc;
            if ((rc = ws.h2d_scalars.ensure(n * SCALAR_BYTES))) return rc;
            HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, scalars, n * SCALAR_BYTES, hipMemcpyHostToDevice, ws.stream));
            const void *dp = nullptr;
            if (points) {
                if ((rc = ws.h2d_points.ensure(n * AFF_BYTES))) return rc;
                HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, points, n * AFF_BYTES, hipMemcpyHostToDevice, ws.stream));
                dp = ws.h2d_points.ptr;
            }
            const SmallPlan splan = small_plan(n, points ? nullptr : resident);
            if ((rc = enqueue_small(ctx, ws, dp, ws.h2d_scalars.ptr, n, splan, resident))) return rc;
            return collect_small(ws, splan, out);
        }
        const WindowPlan plan = plan_for(resident, n);  // the ranges share one bucket set and one reduction
        const unsigned c = plan.c;
        const unsigned nr = host_range_count(n, points != nullptr, plan);
        std::vector<Ext> totals(plan.nwin_total);
        int rc = window_sums_from_host(ctx, first, points, resident, 0, scalars, n, plan, nr, totals.data());
        if (rc) return rc;
        *out = fold(totals.data(), c, plan.nwin_total);
        return GMSM_OK;
    }

    // One rank's piece of a MultiExp sharded over several devices by the library itself (gmsm_multiexp_sharded and the
    // drop-in entries on a multi-GPU node; gmsm_engine.hip): host scalars [0, n) and host points or registered bases
    // [resident_base, resident_base + n), windows win_first, win_first + win_stride, ... of the c-bit decomposition;
    // out_xyzz = the nwin_local window totals. Runs on ctx's device; leases its own workspace.
    static int shard_piece(Context &ctx, const uint64_t *points, const ResidentBases *resident, size_t resident_base,
                           const uint64_t *scalars, size_t n, unsigned c, unsigned win_first, unsigned win_stride,
                           Ext *out_xyzz) {
        WindowPlan plan = make_plan(c, win_first, win_stride);
        // a point slice over bases with window tables of this width: one bucket set (the engine asks for the tables' c)
        if (resident && resident->tab_c.load() == c && win_first == 0 && plan.win_stride == 1 && options().tables.load(std::memory_order_relaxed) != 0)
            plan.shared = 1;
        if (n == 0) {
            for (uint32_t k = 0; k < plan.nwin_local; ++k) out_xyzz[k] = Ext::infinity();
            return GMSM_OK;
        }
        GMSM_LEASE_OR_FAIL(lease, ctx);
        return window_sums_from_host(ctx, *lease.w, points, resident, resident_base, scalars, n, plan,
                                     host_range_count(n, points != nullptr, plan), out_xyzz);
    }

    static int multiexp_host(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                             int nb_tasks, J *out) {
        // argument checks of (*G1Jac).MultiExp, multiexp.go:61-71
        if (n_points != n_scalars) return fail(GMSM_ERR_LEN, "len(points) != len(scalars)");
        if (nb_tasks > 1024) return fail(GMSM_ERR_CONFIG, "invalid config: config.NbTasks > 1024");
        Context *ctx;
        int rc = get_context(&ctx);
        if (rc) return rc;
        HIP_TRY(hipSetDevice(ctx->device));
        const size_t n = n_points;
        if (n == 0) {
            *out = J{F::one(), F::one(), F::zero()};
            return GMSM_OK;
        }
        GMSM_LEASE_OR_FAIL(lease, *ctx);
        return multiexp_from_host(*ctx, *lease.w, points, nullptr, scalars, n, out);
    }
};

}  // namespace gmsm

#include "gmsm_group_debug.h"
#include "gmsm_group_vtable.h"
