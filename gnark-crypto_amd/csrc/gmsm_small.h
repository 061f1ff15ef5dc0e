// The fused small-n MultiExp: ONE launch, one workgroup per (window, slice of SL points), buckets in LDS.
//
// BASELINE.json's config 2 names it ("single-window-per-workgroup Pippenger", LDS-staged buckets per workgroup); the sorted
// pipeline of gmsm_kernels.h replaced it for large n, where throughput counts. Below a few thousand points nothing is
// throughput-bound: the pipeline's fifteen launches are latency chains (0.25 ms at 2^5 points, 0.35 ms at 2^10, of which
// 0.07 ms are the host's Horner fold) while the chip has more lanes than the MSM has additions. The reference benches from
// 2^5 points (multiexp_test.go:344), Pedersen commitments and KZG batch verification issue such sizes (fr/pedersen/
// pedersen.go:100-131, NbTasks: 1). What counts here is the DEPTH of the dependency chain, so every phase is a tree:
//
//   1. digits      thread t decomposes scalar p0 + t for this workgroup's window (partitionScalars, multiexp.go:709-803:
//                  the borrow chain of the windows below is replayed - a few instructions per window)
//   2. grouping    counting sort of the <= SL references by bucket inside LDS (counters, one-wave scan, placement)
//   3. buckets     thread t starts from sorted entry t (+ entry t + 1 when it shares the bucket: mixed additions,
//                  g1.go:822-930), then log2(longest run) - 1 doubling steps E_t += E_{t + d} inside the runs of equal
//                  buckets: the first entry of every run ends up holding its bucket's sum (processChunk's bucket loop,
//                  multiexp_jacobian.go:26-39, as a segmented scan)
//   4. reduction   sum_k (k + 1) B_k = sum of all suffix sums of B (multiexp_jacobian.go:44-52 is the serial form): a suffix
//                  scan over the 2^(c-1) buckets and a tree, 2 (c - 1) steps, no doublings
//   5. slices      n > SL: the workgroup that finishes a window's last slice adds the slices' totals (tree) - the
//                  reference's split of a MultiExp in halves, multiexp.go:98-140
//
// Every step is "load both operands from LDS - barrier - add - store - barrier": one-lane additions where every lane has
// one (phase 3: 8-10 us per step, measured), lane quads where few elements are left (phases 4 and 5: about 4 us).
// Depth for BN254 G1, 1024 points, c = 6 (4 slices of 256): 2 mixed additions + 3 one-lane steps + 10 + 2 quad steps.
// The host folds the window totals as ever (Group::fold: (nwin - 1) c doublings, 0.07 ms for BN254 G1 - the floor of every
// MultiExp that takes its bases anew, whatever the device does).
#pragma once
#include "gmsm_kernels.h"

namespace gmsm {

constexpr uint32_t SMALL_MAX_C = 7;                         // at most 2^6 buckets per window: one lane quad each
constexpr uint32_t SMALL_NB_MAX = 1u << (SMALL_MAX_C - 1);
constexpr uint32_t SMALL_MAX_SLICES = 64;          // slices per window of the plain form
constexpr uint32_t SMALL_SHARED_MAX_SLICES = 1024;  // slices of the shared form (one bucket set: nwin * n entries)

// points per workgroup (= threads): 256 = one wave per SIMD. A step of the kernel is one addition per lane, and two waves
// on a SIMD take turns at its issue port: with 512 points per workgroup every step of phase 3 took twice as long (measured:
// 177-225 us per launch at 2^10 points) while half of the chip's CUs had no workgroup at all.
template <class U> struct SmallSlice { static constexpr uint32_t value = 256u; };

// grid = (nslices, nwin), block = SL, dynamic LDS = SL * sizeof(XYZZL<U>).
// points: Go-layout affine bases (device) or nullptr when upoints (+ skip) are the registered, rewritten bases.
// slice_sums: [nwin][nslices] lazy records, done: [nwin] counters (zero before the launch; the kernel leaves them zero),
// totals: [nwin] canonical XYZZ.
//
// SHARED (narrow window tables of registered bases, Group::precompute_tables): `upoints` holds slab w = 2^(c w) P_i for
// every window, tab_m points per slab; the entries are the nwin * n (window, point) pairs, e = w n + i, grid = (slices, 1);
// every workgroup reduces its own bucket set and the last one adds all slice totals: ONE total, no host-side fold.
template <class U, class FrP, uint32_t SL, bool SHARED>
__global__ void __launch_bounds__(SL) k_msm_small(const void *__restrict__ points, const void *__restrict__ upoints,
                                                  const uint8_t *__restrict__ skip, const uint32_t *__restrict__ scalars,
                                                  uint32_t n, WindowPlan plan, uint32_t tab_m, void *__restrict__ slice_sums,
                                                  uint32_t *__restrict__ done, void *__restrict__ totals) {
    using T = LzTraits<U>;
    using A = UnsatOps<U>;
    using E = typename A::Elem;
    constexpr int NR = FrP::N;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    char *R = reinterpret_cast<char *>(lds_raw);  // SL records (phase 3), reused for the buckets / slices afterwards
    __shared__ uint32_t cnt[SMALL_NB_MAX], pos[SMALL_NB_MAX], start[SMALL_NB_MAX + 1];
    __shared__ uint16_t sorted[SL], sbkt[SL];
    __shared__ uint32_t s_maxrun, s_last;
    const uint32_t t = threadIdx.x, slice = blockIdx.x, nslices = gridDim.x;
    const uint32_t NB = plan.nbuckets;
    const uint32_t e_mine = slice * SL + t;                  // entry of this thread
    const uint32_t w = SHARED ? e_mine / n : blockIdx.y;     // its window (SHARED: per thread) ...
    const uint32_t i = SHARED ? e_mine - w * n : e_mine;     // ... and point
    const uint32_t wout = SHARED ? 0u : blockIdx.y;          // row of the totals / slice sums / counters this workgroup feeds
    const bool have = SHARED ? e_mine < n * plan.nwin_total : i < n;
    if (t < NB) cnt[t] = 0;
    if (t == 0) s_maxrun = 0;
    __syncthreads();

    // ---- 1. the digit of scalar i in window w
    uint32_t code = 0;
    if (have) {
        Fp<FrP> s;
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(scalars + (size_t)i * NR);
            uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
            for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
        }
        bool zero = s.is_zero();                              // multiexp.go:743
        if (skip != nullptr) zero = zero || skip[i] != 0;     // registered bases: infinity flags of the rewrite
        if (!zero) {
            s = fp_from_mont(s);
            const uint32_t c = plan.c, mask = (1u << c) - 1u;
            const int max = (1 << (c - 1)) - 1;
            int carry = 0;
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
        }
        if (code != 0 && points != nullptr) {  // affine (0, 0) contributes nothing (g1.go:825)
            const Affine<typename T::Sat> a = load_struct<Affine<typename T::Sat>>(points, i);
            if (a.is_infinity()) code = 0;
        }
    }
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
    auto fetch = [&](uint32_t ref, U &px, U &py) {
        uint32_t idx = slice * SL + (ref >> 1);
        if constexpr (SHARED) {  // entry -> (window, point) -> slot of the table
            const uint32_t we = idx / n;
            idx = we * tab_m + (idx - we * n);
        }
        if (points != nullptr) {
            const Affine<typename T::Sat> a = load_struct<Affine<typename T::Sat>>(points, idx);
            UAffine<U> u;  // the class the registered bases are stored in (k_convert_points)
            T::pack(T::template from_sat<true>(a.x), u.x);
            T::pack(T::template from_sat<true>(a.y), u.y);
            px = T::unpack(u.x);
            py = T::unpack(u.y);
        } else {
            const UAffine<U> u = load_struct<UAffine<U>>(upoints, idx);
            px = T::unpack(u.x);
            py = T::unpack(u.y);
        }
    };
    if (t < m) {
        E e;
        e.inf = true;
        U px, py;
        const uint32_t r0 = sorted[t];
        fetch(r0, px, py);
        lz_madd_acc<true>(e.v, e.inf, px, py, (r0 & 1u) != 0);
        if (t + 1 < m && sbkt[t + 1] == sbkt[t]) {
            const uint32_t r1 = sorted[t + 1];
            fetch(r1, px, py);
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

    // ---- 4 / 5 on lane quads (gmsm_quad.h): few elements, many idle lanes - a quad addition takes about 4 us where the
    // one-lane addition takes 8-10. The bucket sums (first entry of every run) move into quad records S[0 .. NB), NB <= 64
    // quads; suffix scan, tree; a second round, by the last workgroup of a window only, is the tree over the slices' totals.
    QRec<U> *S = reinterpret_cast<QRec<U> *>(lds_raw);  // aliases R: the move goes through registers
    const uint32_t j = t >> 2, lane = t & 63u;
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
            if (t < 4) {  // quad 0 converts the total: canonical saturated XYZZ for the host (as k_reduce2_q)
                using Mem = XYZZ<typename T::Sat>;
                typename T::Sat *dst = reinterpret_cast<typename T::Sat *>(reinterpret_cast<char *>(totals) + (size_t)wout * sizeof(Mem));
                typename T::Sat v = T::template to_sat<true>(S[0].c[t]);
                if (S[0].inf) {
                    const Mem inf = Mem::infinity();
                    v = t == 0 ? inf.x : t == 1 ? inf.y : t == 2 ? inf.zz : inf.zzz;
                }
                dst[t] = v;
            }
            return;
        }
        if (j == 0) quad_rec_store<U>(slice_sums, (size_t)wout * nslices + slice, &S[0], lane);
        __syncthreads();
        if (t == 0) {
            __threadfence();  // the slice's total is visible device-wide before the counter says so
            const uint32_t seen = __hip_atomic_fetch_add(&done[wout], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = seen + 1u == nslices ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        if (t == 0) done[wout] = 0;  // re-armed for the next call on this workspace
        // the slices' totals: quad j adds up the records j, j + 64, ... (plain form: at most 64 slices, one each), then the tree
        QRec<U> *Tq = S + 64;
        if ((t & 3u) == 0) S[j].inf = 1u;
        __syncthreads();
        for (uint32_t b0 = 0; b0 < nslices; b0 += 64) {
            const uint32_t idx = b0 + j;
            quad_rec_load<U>(&Tq[j], slice_sums, (size_t)wout * nslices + (idx < nslices ? idx : 0), idx < nslices, lane);
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

}  // namespace gmsm
