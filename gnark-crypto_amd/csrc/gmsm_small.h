// The fused small-n MultiExp: ONE launch, one workgroup per (window, slice of the entries), buckets in LDS.
//
// BASELINE.json's config 2 names it ("single-window-per-workgroup Pippenger", LDS-staged buckets per workgroup); the sorted
// pipeline of gmsm_kernels.h replaced it for large n, where throughput counts. Below a few thousand points nothing is
// throughput-bound: the pipeline's fifteen launches are latency chains (0.25 ms at 2^5 points, 0.35 ms at 2^10, of which
// 0.07 ms are the host's Horner fold) while the chip has more lanes than the MSM has additions. The reference benches from
// 2^5 points (multiexp_test.go:344), Pedersen commitments and KZG batch verification issue such sizes (fr/pedersen/
// pedersen.go:100-131, NbTasks: 1). What counts here is the DEPTH of the dependency chain, so every phase is a tree:
//
//   1. digits      one thread per entry decomposes its scalar for this workgroup's window (partitionScalars, multiexp.go:
//                  709-803: the borrow chain of the windows below is replayed - a few instructions per window).
//                  Round 6, GLV (gmsm_glv.h): an entry is a HALF scalar - entry i = (P_i, k1_i), entry n + i = (phi(P_i),
//                  k2_i) - so a window has 2 n entries and there are half as many windows: half the host fold.
//   2. grouping    counting sort of the workgroup's references by bucket inside LDS (counters, one-wave scan, placement)
//   3. buckets     a segmented scan over the sorted entries: E_t += E_{t + d} inside the runs of equal buckets, the first
//                  entry of every run ends up holding its bucket's sum (processChunk's bucket loop, multiexp_jacobian.go:
//                  26-39). Two forms: k_msm_small (one lane per entry, 256 entries per workgroup: mixed additions for the
//                  first step, then one-lane additions) and k_msm_small_q (round 6: one lane QUAD per entry, 64 entries per
//                  chunk, every step a quad addition of gmsm_quad.h - a quarter of the latency of a one-lane step at the
//                  same work, and a third of the registers: the only form the wide element types are built with)
//   4. reduction   sum_k (k + 1) B_k = sum of all suffix sums of B (multiexp_jacobian.go:44-52 is the serial form): a suffix
//                  scan over the 2^(c-1) buckets and a tree, 2 (c - 1) quad steps, no doublings
//   5. slices      more entries than one workgroup takes: the workgroup that finishes a window's last slice adds the slices'
//                  totals (tree) - the reference's split of a MultiExp in halves, multiexp.go:98-140
//
// Every step is "load both operands from LDS - barrier - add - store - barrier".
// The host folds the window totals as ever (Group::fold: (nwin - 1) c doublings - the floor of every MultiExp that takes its
// bases anew, whatever the device does; GLV halves it).
#pragma once
#include "gmsm_glv.h"
#include "gmsm_kernels.h"

namespace gmsm {

constexpr uint32_t SMALL_MAX_C = 7;                         // at most 2^6 buckets per window: one lane quad each
constexpr uint32_t SMALL_NB_MAX = 1u << (SMALL_MAX_C - 1);
constexpr uint32_t SMALL_MAX_SLICES = 64;          // slices per window of the plain form
constexpr uint32_t SMALL_SHARED_MAX_SLICES = 1024;  // slices of the shared form (one bucket set: nwin * n entries)
constexpr uint32_t SMALL_QUAD_ENTRIES = 64;         // entries per chunk of the quad form: one quad each on 256 threads
constexpr uint32_t SMALL_QUAD_MAX_CHUNKS = 8;       // chunks a workgroup of the quad form walks through

// points per workgroup (= threads) of the one-lane form: 256 = one wave per SIMD. A step of the kernel is one addition per
// lane, and two waves on a SIMD take turns at its issue port: with 512 points per workgroup every step of phase 3 took twice
// as long (measured: 177-225 us per launch at 2^10 points) while half of the chip's CUs had no workgroup at all.
template <class U> struct SmallSlice { static constexpr uint32_t value = 256u; };

// The wide element types (Fp2 and the 28-limb field) are built with the quad form only: their one-lane additions need
// 332-512 registers (6-16 of them spilled in round 5) where a quad lane needs a third.
template <class U> struct SmallQuadOnly { static constexpr bool value = sizeof(U) > 14 * 4; };

// Everything the two kernels are told about the call.
struct SmallArgs {
    const void *points;       // Go-layout affine bases (device) or nullptr when upoints (+ skip) are the registered, rewritten bases
    const void *upoints;
    const uint8_t *skip;
    const uint32_t *scalars;
    uint32_t n;               // points / scalars of the call
    uint32_t glv;             // 1: entries are half scalars (2 n per window, plan = the half scalars' windows)
    uint32_t tab_m;           // SHARED: points per slab of the narrow tables
    uint32_t chunks;          // quad form: chunks of 64 entries per workgroup
    void *slice_sums;         // [nwin][nslices] lazy records
    uint32_t *done;           // [nwin] counters, zero before the launch
    void *totals;             // [nwin] canonical XYZZ
};

// ---- 1. the digit code of scalar i (half `half` under GLV) in window w; 0 = contributes nothing
template <class FrP>
__device__ __forceinline__ uint32_t small_code(const SmallArgs &a, const WindowPlan &plan, uint32_t i, uint32_t half, uint32_t w) {
    constexpr int NR = FrP::N, HL = FrP::GLV_HL;
    Fp<FrP> s;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.scalars + (size_t)i * NR);
        uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
    }
    bool zero = s.is_zero();                                  // multiexp.go:743
    if (a.skip != nullptr) zero = zero || a.skip[i] != 0;     // registered bases: infinity flags of the rewrite
    if (zero) return 0u;
    s = fp_from_mont(s);
    bool neg = false;
    if (a.glv) {
        uint32_t k1[HL], k2[HL];
        bool n1, n2;
        glv_split<FrP>(s.l, k1, n1, k2, n2);
        neg = half ? n2 : n1;
#pragma unroll
        for (int k = 0; k < NR; ++k) s.l[k] = k < HL ? (half ? k2[k < HL ? k : 0] : k1[k < HL ? k : 0]) : 0u;
    }
    const uint32_t c = plan.c, mask = (1u << c) - 1u;
    const int max = (1 << (c - 1)) - 1 + (neg ? 1 : 0);  // a negative half: the mirrored digit range (k_decompose_glv)
    int carry = 0;
    uint32_t code = 0;
    for (uint32_t ww = 0; ww <= w; ++ww) {
        const uint32_t bit = ww * c, idx = bit >> 5, sh = bit & 31;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {  // no run-time indexing of the register array
            lo = (uint32_t)k == idx ? s.l[k] : lo;
            hi = (uint32_t)k == idx + 1 ? s.l[k] : hi;
        }
        const uint64_t v = (((uint64_t)hi << 32) | lo) >> sh;
        int digit = carry + (int)((uint32_t)v & mask);
        if (ww + 1 < plan.nwin_total) {
            carry = 0;
            if (digit > max) {
                digit -= 1 << c;
                carry = 1;
            }
            code = digit == 0 ? 0u : (digit > 0 ? ((uint32_t)digit << 1) : ((((uint32_t)(-digit) - 1u) << 1) | 1u));
        } else {
            code = (uint32_t)digit << 1;  // top window: no borrow (multiexp.go:788-800)
        }
    }
    return neg ? code_negate(code) : code;
}

// entry e of a window (plain form) -> point and half
__device__ __forceinline__ void small_entry(const SmallArgs &a, uint32_t e, uint32_t &i, uint32_t &half) {
    half = (a.glv && e >= a.n) ? 1u : 0u;
    i = e - half * a.n;
}

// the affine coordinates of entry e (SHARED: entry -> (window, point) -> slot of the table), lazy domain, packed class
template <class U, class C, bool SHARED>
__device__ __forceinline__ void small_fetch(const SmallArgs &a, uint32_t e, U &px, U &py) {
    using T = LzTraits<U>;
    uint32_t idx = e, half = 0;
    if constexpr (SHARED) {
        const uint32_t we = e / a.n;
        idx = we * a.tab_m + (e - we * a.n);
    } else {
        small_entry(a, e, idx, half);
    }
    if (a.points != nullptr) {
        const Affine<typename T::Sat> p = load_struct<Affine<typename T::Sat>>(a.points, idx);
        UAffine<U> u;  // the class the registered bases are stored in (k_convert_points)
        T::pack(T::template from_sat<true>(p.x), u.x);
        T::pack(T::template from_sat<true>(p.y), u.y);
        px = T::unpack(u.x);
        py = T::unpack(u.y);
    } else {
        const UAffine<U> u = load_struct<UAffine<U>>(a.upoints, idx);
        px = T::unpack(u.x);
        py = T::unpack(u.y);
    }
    if (half) px = glv_mul_w<true>(px, glv_w<U, C, true>());  // phi(P) = (w x, y)
}

// the code of entry e, 0 when its point is (0, 0) (g1.go:825)
template <class U, class FrP, bool SHARED>
__device__ __forceinline__ uint32_t small_entry_code(const SmallArgs &a, const WindowPlan &plan, uint32_t e, uint32_t w_plain) {
    using T = LzTraits<U>;
    uint32_t i, half = 0, w = w_plain;
    if constexpr (SHARED) {
        w = e / a.n;
        i = e - w * a.n;
    } else {
        small_entry(a, e, i, half);
    }
    uint32_t code = small_code<FrP>(a, plan, i, half, w);
    if (code != 0 && a.points != nullptr) {
        const Affine<typename T::Sat> p = load_struct<Affine<typename T::Sat>>(a.points, i);
        if (p.is_infinity()) code = 0;
    }
    return code;
}

// ---- 4 / 5 on lane quads: the bucket sums S[0 .. NB) (NB <= 64 quads) -> suffix scan, tree; a second round, by the last
// workgroup of a window only, is the tree over the slices' totals. S and Tq: 64 QRec each; 256 threads.
template <class U>
__device__ __forceinline__ void small_reduce_and_slices(QRec<U> *S, QRec<U> *Tq, const SmallArgs &a, uint32_t NB, uint32_t wout,
                                                        uint32_t slice, uint32_t nslices, uint32_t *s_last) {
    using T = LzTraits<U>;
    const uint32_t t = threadIdx.x, j = t >> 2, lane = t & 63u;
    uint32_t width = NB;
    bool scan = true;
    for (;;) {
        if (scan) {
#pragma nounroll
            for (uint32_t d = 1; d < width; d <<= 1) {  // inclusive suffix scan: S_j = sum of the buckets j ..
                const bool act = j + d < width;
                const QAddOps<U> o = quad_add_load<U>(&S[j], &S[act ? j + d : j], lane);
                __syncthreads();
                quad_add_store<U, true>(&S[j], o, act, lane);
                __syncthreads();
            }
        }
#pragma nounroll
        for (uint32_t d = width >> 1; d >= 1; d >>= 1) {  // tree: the sum of all suffix sums = sum_k (k + 1) B_k
            const bool act = j < d;
            const QAddOps<U> o = quad_add_load<U>(&S[j], &S[act ? j + d : j], lane);
            __syncthreads();
            quad_add_store<U, true>(&S[j], o, act, lane);
            __syncthreads();
        }
        // S[0] = this round's result
        if (nslices == 1 || !scan) {
            if (t < 4) {  // quad 0 converts the total: canonical saturated XYZZ for the host, (1, 1, 0, 0) for infinity (g1.go:688)
                using Sat = typename T::Sat;
                Sat *dst = reinterpret_cast<Sat *>(reinterpret_cast<char *>(a.totals) + (size_t)wout * sizeof(XYZZ<Sat>));
                Sat v = T::template to_sat<true>(S[0].c[t]);
                if (S[0].inf) v = t < 2 ? Sat::one() : Sat::zero();
                dst[t] = v;
            }
            return;
        }
        if (j == 0) quad_rec_store<U>(a.slice_sums, (size_t)wout * nslices + slice, &S[0], lane);
        __syncthreads();
        if (t == 0) {
            __threadfence();  // the slice's total is visible device-wide before the counter says so
            const uint32_t seen = __hip_atomic_fetch_add(&a.done[wout], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            *s_last = seen + 1u == nslices ? 1u : 0u;
        }
        __syncthreads();
        if (!*s_last) return;
        __threadfence();
        if (t == 0) a.done[wout] = 0;  // re-armed (the host also clears the counters before every multi-slice launch)
        // the slices' totals: quad j adds up the records j, j + 64, ... (plain form: at most 64 slices, one each), then the tree
        if ((t & 3u) == 0) S[j].inf = 1u;
        __syncthreads();
        for (uint32_t b0 = 0; b0 < nslices; b0 += 64) {
            const uint32_t idx = b0 + j;
            quad_rec_load<U>(&Tq[j], a.slice_sums, (size_t)wout * nslices + (idx < nslices ? idx : 0), idx < nslices, lane);
            __syncthreads();
            const QAddOps<U> o = quad_add_load<U>(&S[j], &Tq[j], lane);
            __syncthreads();
            quad_add_store<U, true>(&S[j], o, idx < nslices, lane);
            __syncthreads();
        }
        width = 1;
        while (width < nslices && width < 64) width <<= 1;
        scan = false;
    }
}

// ------------------------------------------------------------------ the one-lane form
// grid = (nslices, nwin), block = SL, dynamic LDS = SL * sizeof(XYZZL<U>) (>= 128 QRec<U>).
//
// SHARED (narrow window tables of registered bases, Group::precompute_tables): `upoints` holds slab w = 2^(c w) P_i for
// every window, tab_m points per slab; the entries are the nwin * n (window, point) pairs, e = w n + i, grid = (slices, 1);
// every workgroup reduces its own bucket set and the last one adds all slice totals: ONE total, no host-side fold.
template <class U, class FrP, class C, uint32_t SL, bool SHARED>
__global__ void __launch_bounds__(SL) k_msm_small(SmallArgs a, WindowPlan plan) {
    using A = UnsatOps<U>;
    using E = typename A::Elem;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    char *R = reinterpret_cast<char *>(lds_raw);  // SL records (phase 3), reused for the buckets / slices afterwards
    __shared__ uint32_t cnt[SMALL_NB_MAX], pos[SMALL_NB_MAX], start[SMALL_NB_MAX + 1];
    __shared__ uint16_t sorted[SL], sbkt[SL];
    __shared__ uint32_t s_maxrun, s_last;
    const uint32_t t = threadIdx.x, slice = blockIdx.x, nslices = gridDim.x;
    const uint32_t NB = plan.nbuckets;
    const uint32_t ne = SHARED ? a.n * plan.nwin_total : (a.glv ? 2u * a.n : a.n);  // entries of a bucket set
    const uint32_t e_mine = slice * SL + t;                                        // entry of this thread
    const uint32_t wout = SHARED ? 0u : blockIdx.y;          // row of the totals / slice sums / counters this workgroup feeds
    if (t < NB) cnt[t] = 0;
    if (t == 0) s_maxrun = 0;
    __syncthreads();

    const uint32_t code = e_mine < ne ? small_entry_code<U, FrP, SHARED>(a, plan, e_mine, blockIdx.y) : 0u;
    // ---- 2. counting sort of the references by bucket
    const uint32_t b = code ? code_bucket(code) : 0u;
    if (code) atomicAdd(&cnt[b], 1u);
    __syncthreads();
    if (t < NB) atomicMax(&s_maxrun, cnt[t]);
    {
        const uint32_t tot = wave0_exclusive_scan(cnt, NB, 0u, start, pos);
        if (t == 0) start[NB] = tot;
    }
    __syncthreads();
    if (code) {
        const uint32_t p = atomicAdd(&pos[b], 1u);
        sorted[p] = (uint16_t)((t << 1) | (code & 1u));
        sbkt[p] = (uint16_t)b;
    }
    __syncthreads();
    const uint32_t m = start[NB], maxrun = s_maxrun;

    // ---- 3a. E_t = P(sorted[t]) (+ P(sorted[t + 1]) inside the same bucket)
    if (t < m) {
        E e;
        e.inf = true;
        U px, py;
        const uint32_t r0 = sorted[t];
        small_fetch<U, C, SHARED>(a, slice * SL + (r0 >> 1), px, py);
        lz_madd_acc<true>(e.v, e.inf, px, py, (r0 & 1u) != 0);
        if (t + 1 < m && sbkt[t + 1] == sbkt[t]) {
            const uint32_t r1 = sorted[t + 1];
            small_fetch<U, C, SHARED>(a, slice * SL + (r1 >> 1), px, py);
            lz_madd_acc<true>(e.v, e.inf, px, py, (r1 & 1u) != 0);
        }
        lz_acc_finish(e.v, e.inf);
        A::store(R, t, e);
    }
    __syncthreads();

    // ---- 3b. doubling steps inside the runs: E_t += E_{t + d}, d = 2, 4, ... < longest run (one-lane additions: every
    // lane has one)
    uint32_t n3 = 0;
    while ((2u << n3) < maxrun) ++n3;
#pragma nounroll
    for (uint32_t s = 0; s < n3; ++s) {
        const uint32_t src = t + (2u << s);
        const bool act = src < m && sbkt[src] == sbkt[t];
        E p = A::infinity(), q = A::infinity();
        if (act) {
            p = A::load(R, t);
            q = A::load(R, src);
        }
        __syncthreads();
        if (act) {
            A::add(p, q);
            A::store(R, t, p);
        }
        __syncthreads();
    }

    // ---- 4 / 5 on lane quads. The bucket sums (first entry of every run) move into quad records S[0 .. NB); S aliases R:
    // the move goes through registers.
    QRec<U> *S = reinterpret_cast<QRec<U> *>(lds_raw);
    {
        E bsum = A::infinity();
        if (t < NB && start[t + 1] > start[t]) bsum = A::load(R, start[t]);
        __syncthreads();
        if (t < NB) {
            S[t].c[0] = bsum.v.x;
            S[t].c[1] = bsum.v.y;
            S[t].c[2] = bsum.v.zz;
            S[t].c[3] = bsum.v.zzz;
            S[t].inf = bsum.inf ? 1u : 0u;
        }
        __syncthreads();
    }
    small_reduce_and_slices<U>(S, S + 64, a, NB, wout, slice, nslices, &s_last);
}

// ------------------------------------------------------------------ the quad form
// grid = (nslices, nwin), block = 256 = 64 lane quads, dynamic LDS = 192 * sizeof(QRec<U>): R (the chunk's entries), S (the
// bucket sums of the chunks so far), Tq (the slices' totals in phase 5). A workgroup walks a.chunks chunks of 64 entries:
// digits - sort - one record per entry (affine coordinates, zz = zzz = 1) - segmented scan on quads (d = 1, 2, 4, ... < longest
// run) - S[b] += head of run b.
template <class P>
__device__ __forceinline__ FpU<P> small_neg_y_impl(const FpU<P> &y) { return fpu_neg4<P>(y); }                       // < 6: the class of a stored y
template <class P>
__device__ __forceinline__ Fp2U<P> small_neg_y_impl(const Fp2U<P> &y) { return lz_sub(lz_zero((const Fp2U<P> *)nullptr), y); }  // class R

template <class U, class FrP, class C, bool SHARED>
__global__ void __launch_bounds__(256) k_msm_small_q(SmallArgs a, WindowPlan plan) {
    constexpr uint32_t EQ = SMALL_QUAD_ENTRIES;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *R = reinterpret_cast<QRec<U> *>(lds_raw), *S = R + EQ, *Tq = S + 64;
    __shared__ uint32_t cnt[SMALL_NB_MAX], pos[SMALL_NB_MAX], start[SMALL_NB_MAX + 1];
    __shared__ uint16_t sorted[EQ], sbkt[EQ];
    __shared__ uint32_t s_maxrun, s_last;
    const uint32_t t = threadIdx.x, j = t >> 2, r = t & 3u, lane = t & 63u;
    const uint32_t slice = blockIdx.x, nslices = gridDim.x;
    const uint32_t NB = plan.nbuckets;
    const uint32_t ne = SHARED ? a.n * plan.nwin_total : (a.glv ? 2u * a.n : a.n);
    const uint32_t wout = SHARED ? 0u : blockIdx.y;
    if (r == 0u && j < 64u) S[j].inf = 1u;
#pragma nounroll
    for (uint32_t ch = 0; ch < a.chunks; ++ch) {
        const uint32_t e0 = (slice * a.chunks + ch) * EQ;
        if (e0 >= ne) break;  // uniform
        if (t < NB) cnt[t] = 0;
        if (t == 0) s_maxrun = 0;
        __syncthreads();
        uint32_t code = 0;
        if (t < EQ && e0 + t < ne) code = small_entry_code<U, FrP, SHARED>(a, plan, e0 + t, blockIdx.y);
        const uint32_t b = code ? code_bucket(code) : 0u;
        if (code) atomicAdd(&cnt[b], 1u);
        __syncthreads();
        if (t < NB) atomicMax(&s_maxrun, cnt[t]);
        {
            const uint32_t tot = wave0_exclusive_scan(cnt, NB, 0u, start, pos);
            if (t == 0) start[NB] = tot;
        }
        __syncthreads();
        if (code) {
            const uint32_t p = atomicAdd(&pos[b], 1u);
            sorted[p] = (uint16_t)((t << 1) | (code & 1u));
            sbkt[p] = (uint16_t)b;
        }
        __syncthreads();
        const uint32_t m = start[NB], maxrun = s_maxrun;
        // ---- 3a. quad j holds entry sorted[j]: lane 0 x, lane 1 (+-) y, lanes 2, 3 the ones
        if (j < m) {
            const uint32_t ref = sorted[j];
            if (r < 2u) {
                U px, py;
                small_fetch<U, C, SHARED>(a, e0 + (ref >> 1), px, py);
                if (r == 0u) {
                    R[j].c[0] = px;
                    R[j].inf = 0u;
                } else {
                    R[j].c[1] = (ref & 1u) ? small_neg_y_impl(py) : py;
                }
            } else {
                R[j].c[r] = lz_one((const U *)nullptr);
            }
        }
        __syncthreads();
        // ---- 3b. segmented scan on quads
#pragma nounroll
        for (uint32_t d = 1; d < maxrun; d <<= 1) {
            const bool act = j < m && j + d < m && sbkt[j + d] == sbkt[j];
            const QAddOps<U> o = quad_add_load<U>(&R[j < m ? j : 0], &R[act ? j + d : (j < m ? j : 0)], lane);
            __syncthreads();
            quad_add_store<U, true>(&R[j < m ? j : 0], o, act, lane);
            __syncthreads();
        }
        // ---- S[b] += head of run b (quad b)
        {
            const bool act = j < NB && start[j + 1] > start[j];
            const QAddOps<U> o = quad_add_load<U>(&S[j], &R[act ? start[j] : 0], lane);
            __syncthreads();
            quad_add_store<U, true>(&S[j], o, act, lane);
            __syncthreads();
        }
    }
    __syncthreads();
    small_reduce_and_slices<U>(S, Tq, a, NB, wout, slice, nslices, &s_last);
}

}  // namespace gmsm
