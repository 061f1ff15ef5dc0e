// HIP kernels of the Pippenger pipeline (gfx950). One template per stage, instantiated per (curve, group) in
// gmsm_group_inst.hip. Stage -> reference function it replaces:
//
//   k_decompose        partitionScalars                    ecc/bn254/multiexp.go:709-803 (+ fr.Bits/fromMont, fr/element.go:855)
//   k_part_hist/k_part_colscan/k_part_rowscan/k_part_scatter/k_fine_sort
//                      (no reference twin) group each window's point references by bucket, so that
//                      every bucket is owned by one accumulation thread at a time -- replaces the reference's
//                      "one goroutine walks all n digits of a window" (multiexp_jacobian.go:26-39)
//   k_convert_points   rewrite the bases into the lazy Montgomery domain (once per call or per registered SRS)
//   k_accumulate_seg   bucket accumulation loop             multiexp_jacobian.go:26-39 (addMixed / subMixed)
//   k_fixup_seg/long   close buckets that were split over several accumulation threads
//   k_reduce_serial[_q], k_combine_q / k_combine_we, k_reduce2_q (gmsm_quad.h)
//                      running-sum bucket reduction         multiexp_jacobian.go:44-52
//   k_merge_buckets    the split of a MultiExp in point ranges  multiexp.go:98-140 (AddAssign, here bucket by bucket)
//   (host) fold        msmReduceChunkG1Affine               multiexp.go:302-315
//
// Digit code (the reference's uint16 digits, multiexp.go:779-800; stored as uint16 whenever every code of the call fits,
// i.e. for all c <= 16 in scope, else uint32 - the digit arrays are the largest intermediate of the front end):
//   0 = skip, d > 0 -> 2d, d < 0 -> 2(-d-1)+1;  bucket = (code>>1) - ((code&1)^1), negate = code&1.
// Sorted entry: (point_index << 1) | negate.
#pragma once
#include <hip/hip_runtime.h>
#include "gmsm_curveu.h"
#include "gmsm_quad.h"

namespace gmsm {

// Window geometry shared by host and device.
struct WindowPlan {
    uint32_t c;          // window width in bits
    uint32_t nwin_total; // number of windows of the full scalar (= computeNbChunks(c), multiexp.go:681)
    uint32_t nbuckets;   // buckets allocated per window: 2^(max(c,lastC)-1)
    uint32_t win_first;  // this launch handles windows win_first + k*win_stride, k < nwin_local (window sharding)
    uint32_t win_stride;
    uint32_t nwin_local;
    uint32_t shared;     // 1: window tables (Group::precompute_tables) - the digits of all windows index ONE bucket set,
                         // entry (w, i) gathers 2^(c w) P_i; the launch reports the total as window 0 and infinity above
    uint32_t glv;        // 1: the windows are those of the GLV half scalars (gmsm_glv.h): nwin_total = ceil(GLV_BITS / c), a window
                         // has 2 n entries - entry 2 i = (P_i, k1_i), entry 2 i + 1 = (phi(P_i), k2_i) (the fused small-n kernel numbers them i and n + i)
};

template <class T>
__device__ __forceinline__ T load_struct(const void *base, size_t index) {
    // sizeof(T) is a multiple of 16 for every element type in scope (32..384 B): 16-byte vector loads
    static_assert(sizeof(T) % 16 == 0, "element size");
    T r;
    const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(base) + index * sizeof(T));
    uint4 *dst = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 16); ++i) dst[i] = src[i];
    return r;
}

template <class T>
__device__ __forceinline__ void store_struct(void *base, size_t index, const T &v) {
    static_assert(sizeof(T) % 16 == 0, "element size");
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(base) + index * sizeof(T));
    const uint4 *src = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 16); ++i) dst[i] = src[i];
}

// ------------------------------------------------------------------ scalar decomposition
// One thread per scalar. digits is [nwin_local][n] (window-major, coalesced stores).
template <class FrP, class D>
__global__ void __launch_bounds__(256) k_decompose(const uint32_t *__restrict__ scalars, size_t n, WindowPlan plan,
                                                   D *__restrict__ digits,
                                                   const uint8_t *__restrict__ skip /* may be null */) {
    constexpr int NR = FrP::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrP> s;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scalars + i * NR);
        uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
    }
    // zero scalars (multiexp.go:743) and points at infinity (g1.go:825) contribute nothing
    const bool zero = s.is_zero() || (skip != nullptr && skip[i] != 0);
    s = fp_from_mont(s);
    const uint32_t c = plan.c;
    const uint32_t mask = (1u << c) - 1u;
    const int max = (1 << (c - 1)) - 1;
    int carry = 0;
    for (uint32_t w = 0; w < plan.nwin_total; ++w) {
        const uint32_t bit = w * c, idx = bit >> 5, sh = bit & 31;
        // up to 3 words can contribute when c > 32-sh ... c <= 24 is enforced on the host: 2 words suffice
        // for c+sh <= 64
        uint64_t lo = s.l[idx];
        uint64_t hi = (idx + 1 < (uint32_t)NR) ? s.l[idx + 1] : 0u;
        uint64_t v = ((hi << 32) | lo) >> sh;
        int digit = carry + (int)((uint32_t)v & mask);
        uint32_t code;
        if (w + 1 < plan.nwin_total) {
            carry = 0;
            if (digit > max) {
                digit -= 1 << c;
                carry = 1;
            }
            code = digit == 0 ? 0u : (digit > 0 ? ((uint32_t)digit << 1) : ((((uint32_t)(-digit) - 1u) << 1) | 1u));
        } else {
            code = (uint32_t)digit << 1;  // top window: no borrow (multiexp.go:788-800)
        }
        if (w >= plan.win_first && (w - plan.win_first) % plan.win_stride == 0) {
            const uint32_t k = (w - plan.win_first) / plan.win_stride;
            if (k < plan.nwin_local) digits[(size_t)k * n + i] = (D)(zero ? 0u : code);
        }
    }
}

// The same with the window width as a template parameter: the loop over the windows unrolls, every limb index and shift
// is a constant and the scalar stays in registers (the generic kernel indexes its limbs with a run-time value, which the
// compiler serves from a 12 KB LDS array: 44 us for 2^20 scalars where the traffic needs 15). Instantiated for the widths
// of the window table (preferred_c); forced or test widths take the generic kernel. Same digits, bit for bit.
template <class FrP, class D, int C>
__global__ void __launch_bounds__(256) k_decompose_c(const uint32_t *__restrict__ scalars, size_t n, WindowPlan plan,
                                                     D *__restrict__ digits, const uint8_t *__restrict__ skip) {
    constexpr int NR = FrP::N;
    constexpr uint32_t NW = (FrP::BITS + C - 1) / C;  // computeNbChunks
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<FrP> s;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scalars + i * NR);
        uint4 *dst = reinterpret_cast<uint4 *>(s.l);
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) dst[k] = src[k];
    }
    const bool zero = s.is_zero() || (skip != nullptr && skip[i] != 0);
    s = fp_from_mont(s);
    constexpr uint32_t mask = (1u << C) - 1u;
    constexpr int max = (1 << (C - 1)) - 1;
    int carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; ++w) {
        const uint32_t bit = w * C, idx = bit >> 5, sh = bit & 31;
        const uint64_t lo = s.l[idx];
        const uint64_t hi = (idx + 1 < (uint32_t)NR) ? s.l[idx + 1 < (uint32_t)NR ? idx + 1 : idx] : 0u;
        const uint64_t v = ((hi << 32) | lo) >> sh;
        int digit = carry + (int)((uint32_t)v & mask);
        uint32_t code;
        if (w + 1 < NW) {
            carry = 0;
            if (digit > max) {
                digit -= 1 << C;
                carry = 1;
            }
            code = digit == 0 ? 0u : (digit > 0 ? ((uint32_t)digit << 1) : ((((uint32_t)(-digit) - 1u) << 1) | 1u));
        } else {
            code = (uint32_t)digit << 1;  // top window: no borrow (multiexp.go:788-800)
        }
        if (w >= plan.win_first && (w - plan.win_first) % plan.win_stride == 0) {
            const uint32_t k = (w - plan.win_first) / plan.win_stride;
            if (k < plan.nwin_local) digits[(size_t)k * n + i] = (D)(zero ? 0u : code);
        }
    }
}

__device__ __forceinline__ uint32_t code_bucket(uint32_t code) { return (code >> 1) - ((code & 1u) ^ 1u); }

// LDS counters under repeated keys. The counting sorts below keep one counter per partition / bucket in LDS and bump it
// once per entry; lanes of a wave that hit the SAME counter are serialised by the LDS. Uniform scalars hardly do that, the
// reference's own benchmark distributions do (multiexp_test.go:319-334): runs of 100 equal scalars put the same key in 12-25
// adjacent lanes, a scalar repeated n/5 times makes one bucket hold 96 % of its partition, equal scalars all of it. Measured
// (tools/ubench_ldsagg.hip, profiles/r05_ldsagg.log; ns per wave-level update and CU): uniform keys 4.2, runs of 25 lanes 36,
// 96 % one key 52. What helps is aggregating RUNS of equal keys in adjacent lanes - which is how such entries arrive, the
// passes keep index order -: heads = lanes whose key differs from the lane before (row_shr:1, so a 16-lane row always starts
// a run); the head adds the run's length, the other lanes take base + rank through one ds_bpermute: 14 / 15 ns for the two
// patterns above. (Peeling the first lane's key with readlane + ballot rounds, the first form tried, costs more than the
// conflicts it removes for runs of 25 and 60 % extra on uniform keys.) It is not free on uniform keys either (5.6 ns), so
// the callers decide per BATCH of entries from one sample (wave_runny) and use plain atomics otherwise.
constexpr uint32_t AGG_MAX_HEADS = 16;  // a sample with at most this many runs per wave (average run >= 4 lanes) aggregates
__device__ __forceinline__ uint64_t run_heads(uint32_t key, bool active) {
    const uint32_t keyx = active ? key : 0xFFFFFFFFu;  // lanes without an entry form runs of their own
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~keyx, (int)keyx, 0x111, 0xf, 0xf, false);
    return __ballot(prev != keyx);
}
__device__ __forceinline__ bool wave_runny(uint32_t key, bool active) {
    return (uint32_t)__popcll((unsigned long long)run_heads(key, active)) <= AGG_MAX_HEADS;
}
// RET: returns the lane's slot (the counter's value before its own increment)
template <bool RET>
__device__ __forceinline__ uint32_t lds_count_runs(uint32_t *cnt, uint32_t key, bool active) {
    const uint64_t H = run_heads(key, active);
    const uint32_t lane = __lane_id();
    const uint64_t le = ~0ull >> (63u - lane);  // lanes <= this one
    const uint32_t head = 63u - (uint32_t)__clzll((long long)(H & le));
    const uint64_t above = (H | ~__ballot(true)) & ~le;  // a run ends at the next head or at the first lane that is not executing
    const uint32_t next = above ? (uint32_t)__ffsll((long long)above) - 1u : 64u;
    uint32_t base = 0;
    if (lane == head && active) {
        const uint32_t r = atomicAdd(&cnt[key], next - head);
        if constexpr (RET) base = r;
    }
    if constexpr (RET) return (uint32_t)__shfl((int)base, (int)head, 64) + (lane - head);
    return 0u;
}
template <bool RET>
__device__ __forceinline__ uint32_t lds_count(uint32_t *cnt, uint32_t key, bool active, bool runny /* wave-uniform */) {
    if (runny) return lds_count_runs<RET>(cnt, key, active);
    uint32_t r = 0;
    if (active) r = atomicAdd(&cnt[key], 1u);
    return RET ? r : 0u;
}

// ------------------------------------------------------------------ two-level grouping (coarse partition, fine sort)
// A single-pass counting sort would write every 4-byte reference to an effectively random address of the window's
// sorted array: rocprofv3 showed 32 B of HBM write per 4-byte store (r01f: 8.5 GB for 268 M references at 2^24).
// Two-level form: (A) references are first distributed into P coarse partitions (bucket >> fbits) -- a workgroup
// writes one contiguous run per partition; (B) one workgroup per (window, partition) counting-sorts its run by the low
// bucket bits with the cursors in LDS; its scattered writes stay inside a region of a few tens of KB, i.e. they
// coalesce in L2 before reaching HBM. Partition populations are ~16 K references for uniform scalars; any population
// is handled (the fine pass streams its run twice), skewed inputs just lose the locality benefit.
// Entry format between the passes: (fine_bucket << lidx) | (point_index << 1 | negate).

// Exclusive scan of cnt[0..n) in LDS by the first wavefront of the workgroup (lane-serial chunks + a 6-step shuffle scan;
// no workgroup barriers inside). Writes the exclusive prefix (plus `base`) to out[] and returns the total in every lane
// of wave 0. Call from all threads; followed by a __syncthreads() by the caller.
__device__ __forceinline__ uint32_t wave0_exclusive_scan(const uint32_t *cnt, uint32_t n, uint32_t base, uint32_t *out,
                                                         uint32_t *out2 = nullptr) {
    uint32_t total = 0;
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        const uint32_t per = (n + 63) / 64;
        const uint32_t lo = lane * per, hi = lo + per < n ? lo + per : n;
        uint32_t s = 0;
        for (uint32_t i = lo; i < hi; ++i) s += cnt[i];
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += v;
        }
        total = __shfl(incl, 63, 64);
        uint32_t run = base + incl - s;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = cnt[i];
            out[i] = run;
            if (out2) out2[i] = run;
            run += c;
        }
    }
    return total;
}

// grid = (nchunks, nwin). LDS: P counters. blockhist[k][chunk][p]
template <class D>
__global__ void __launch_bounds__(1024) k_part_hist(const D *__restrict__ digits, size_t n, uint32_t nparts,
                                                           uint32_t fbits, size_t chunk_len,
                                                           uint32_t *__restrict__ blockhist) {
    extern __shared__ uint32_t lds_cnt[];
    const uint32_t chunk = blockIdx.x, k = blockIdx.y, nchunks = gridDim.x;
    for (uint32_t p = threadIdx.x; p < nparts; p += blockDim.x) lds_cnt[p] = 0;
    __syncthreads();
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < n ? lo + chunk_len : n;
    const D *d = digits + (size_t)k * n;
    // 16-byte loads (4 or 8 codes per lane and request): with one 2- or 4-byte load per lane and iteration the pass ran at
    // the latency of its loads, not at the rate of HBM or of the LDS atomics (2^24: 0.335 ms for 1 GB, the same with half
    // the bytes - profiles/r04_dig17_ab.log). The order of the codes does not matter to a histogram.
    constexpr size_t PERV = 16 / sizeof(D);
    auto count = [&](uint32_t code, bool runny) { lds_count<false>(lds_cnt, code_bucket(code) >> fbits, code != 0u, runny); };
    size_t a0 = lo + ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(d + lo) & 15u)) & 15u) / sizeof(D);  // first aligned code
    if (a0 > hi) a0 = hi;
    const size_t nvec = (hi - a0) / PERV, a1 = a0 + nvec * PERV;
    for (size_t i = lo + threadIdx.x; i < a0; i += blockDim.x) count(d[i], false);
    const uint4 *dv = reinterpret_cast<const uint4 *>(d + a0);
    for (size_t v = threadIdx.x; v < nvec; v += blockDim.x) {
        const uint4 q = dv[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        const uint32_t c0 = sizeof(D) == 4 ? w[0] : (w[0] & 0xffffu);
        const bool runny = wave_runny(code_bucket(c0) >> fbits, c0 != 0u);  // lanes hold consecutive groups of codes
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (sizeof(D) == 4) {
                count(w[j], runny);
            } else {
                count(w[j] & 0xffffu, runny);
                count(w[j] >> 16, runny);
            }
        }
    }
    for (size_t i = a1 + threadIdx.x; i < hi; i += blockDim.x) count(d[i], false);
    __syncthreads();
    uint32_t *out = blockhist + ((size_t)k * nchunks + chunk) * nparts;
    for (uint32_t p = threadIdx.x; p < nparts; p += blockDim.x) out[p] = lds_cnt[p];
}

// grid = (ceil(nparts/32), nwin), block = 32 partitions x SEGS chunk segments: turns the per-chunk counts of every
// (window, partition) into exclusive prefixes over the chunks (in place) and emits the partition population. Each
// thread sums its segment of the chunks, the SEGS segment sums of a partition are combined through LDS, then the segment
// is rewritten with running prefixes. Adjacent lanes work on adjacent partitions (128-byte rows of blockhist).
// SEGS = 8 when the grid fills the chip; 32 for the few wide columns of a shared bucket set (one window of nwin * n
// entries: thousands of chunks, 16-64 workgroups - a thread's walk over its chunks is what the kernel takes).
template <uint32_t SEGS>
__global__ void __launch_bounds__(32 * SEGS) k_part_colscan(uint32_t *__restrict__ blockhist, uint32_t nchunks, uint32_t nparts,
                                                            uint32_t *__restrict__ part_pop,
                                                            uint32_t *__restrict__ zero_word /* may be null */) {
    __shared__ uint32_t seg_sum[SEGS][32];
    const uint32_t lane = threadIdx.x & 31u, s = threadIdx.x >> 5, k = blockIdx.y;
    if (zero_word != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_word = 0;  // the launch's count of oversized sub-runs (k_part_rowscan adds to it)
    const uint32_t p = blockIdx.x * 32u + lane;
    const uint32_t per = (nchunks + SEGS - 1u) / SEGS;
    const uint32_t c0 = s * per < nchunks ? s * per : nchunks;
    const uint32_t c1 = c0 + per < nchunks ? c0 + per : nchunks;
    uint32_t *bh = blockhist + (size_t)k * nchunks * nparts + p;
    uint32_t mine = 0;
    if (p < nparts)
        for (uint32_t ch = c0; ch < c1; ++ch) mine += bh[(size_t)ch * nparts];
    seg_sum[s][lane] = mine;
    __syncthreads();
    if (p >= nparts) return;
    uint32_t run = 0;
    for (uint32_t q = 0; q < s; ++q) run += seg_sum[q][lane];
    for (uint32_t ch = c0; ch < c1; ++ch) {
        const uint32_t v = bh[(size_t)ch * nparts];
        bh[(size_t)ch * nparts] = run;
        run += v;
    }
    if (s == SEGS - 1u) part_pop[(size_t)k * nparts + p] = run;
}

// Oversized partitions. A coarse partition normally holds ~2^13..2^15 references and one k_fine_sort workgroup sorts it
// inside LDS. A partition that exceeds the staging slots (stage_cap) is "oversized": a bucket that very many scalars
// share (a repeated scalar value fills the same bucket of every window, the value 1 fills bucket 0 of window 0, all
// scalars equal fill one bucket per window with all n references) or a narrow top window (BW6-761: 2^8 buckets of
// n / 2^8 references). Rounds 1-4 sorted such a partition with ONE workgroup, same-address LDS atomics and scattered
// 4-byte stores: 6.1 ms per launch at 2^24 under the reference's "smallvalues" distribution, 29 ms with all scalars equal
// (profiles/r05_distributions_before.log). Now k_part_rowscan lists them, and three kernels sort each one with as many
// workgroups as it has sub-runs of HEAVY_SUB references: k_heavy_hist (per-sub-run bucket counts) -> k_heavy_scan
// (bucket starts + per-sub-run write cursors) -> k_heavy_place (placement; the references of a crowded bucket get
// consecutive slots from one ballot, so their stores coalesce). With no oversized partition the three launches exit on
// their first load.
struct HeavyPart {
    uint32_t p, first, nsub, pad;  // partition, first row of its sub-run counts in the window's table, sub-runs
};
constexpr uint32_t HEAVY_SUB = 8192;

// grid = nwin, block = 1024: exclusive scan of the partition populations -> part_base[k][0..nparts]; lists the window's
// oversized partitions in hparts[k][0..hcount[2k]), hcount[2k + 1] = their sub-runs (hparts == nullptr: none wanted)
static __global__ void __launch_bounds__(1024) k_part_rowscan(const uint32_t *__restrict__ part_pop, uint32_t nparts,
                                                              uint32_t *__restrict__ part_base,
                                                              uint32_t *__restrict__ clear_flag /* [nwin], may be null */,
                                                              uint32_t stage_cap, HeavyPart *__restrict__ hparts,
                                                              uint32_t *__restrict__ hcount, uint32_t hcap) {
    __shared__ uint32_t sums[1024];
    __shared__ uint32_t s_np, s_first;
    const uint32_t k = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    if (clear_flag != nullptr && t == 0) clear_flag[k] = 0;  // the window's long-chain counter (the fix-up kernels raise it)
    if (t == 0) {
        s_np = 0;
        s_first = 0;
    }
    const uint32_t *pp = part_pop + (size_t)k * nparts;
    uint32_t *pb = part_base + (size_t)k * (nparts + 1);
    const uint32_t per = (nparts + T - 1) / T;
    const uint32_t plo = t * per, phi = plo + per < nparts ? plo + per : nparts;
    uint32_t mine = 0;
    for (uint32_t p = plo; p < phi; ++p) mine += pp[p];
    sums[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {
        const uint32_t v = t >= d ? sums[t - d] : 0u;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    uint32_t run = sums[t] - mine;
    for (uint32_t p = plo; p < phi; ++p) {
        const uint32_t pop = pp[p];
        pb[p] = run;
        run += pop;
        if (hparts != nullptr && pop > stage_cap) {
            const uint32_t nsub = (pop + HEAVY_SUB - 1) / HEAVY_SUB;
            const uint32_t j = atomicAdd(&s_np, 1u);
            const uint32_t first = atomicAdd(&s_first, nsub);
            if (j < hcap) hparts[(size_t)k * hcap + j] = HeavyPart{p, first, nsub, 0u};
        }
    }
    if (t == T - 1) pb[nparts] = sums[T - 1];
    if (hparts != nullptr) {
        __syncthreads();
        if (t == 0) {
            hcount[2 * k] = s_np < hcap ? s_np : hcap;
            hcount[2 * k + 1] = s_first;  // sub-runs = rows of the window's table
            if (s_first) atomicAdd(&hcount[2 * gridDim.x], s_first);  // all windows: what the k_heavy_* launches test first
        }
    }
}

// grid = (nchunks, nwin), block = 1024, chunk_len <= CHUNK. The chunk is first sorted by partition inside LDS
// (staging buffer + partition id per slot), then written out slot by slot: consecutive slots of one partition go to
// consecutive addresses, so every (chunk, partition) run leaves the CU as one burst instead of trickling out 4 bytes at
// a time over the lifetime of the workgroup (which left L2 writing back partially filled lines: 4.4 ms at 2^24).
// Chunk = what one workgroup stages in LDS. 12 K entries (76 KB) let two workgroups share a CU - one counts and scans
// while the other places and writes: scatter + fine sort 1.93 -> 1.47 ms at 2^24, 0.42 -> 0.35 at 2^22 (A/B in one
// session, profiles/r03_part_chunk.log). Beyond 2^25 points the coarse pass has 2048 partitions and the longer chunk's
// longer runs win again (2^26: 9.5 against 10.3 ms): PART_CHUNK_BIG.
constexpr uint32_t PART_CHUNK = 12288, PART_CHUNK_BIG = 16384;
template <class D, uint32_t CHUNK>
__global__ void __launch_bounds__(1024) k_part_scatter(const D *__restrict__ digits, size_t n, uint32_t nparts,
                                                              uint32_t fbits, uint32_t lidx, size_t chunk_len,
                                                              const uint32_t *__restrict__ blockhist,
                                                              const uint32_t *__restrict__ part_base,
                                                              uint32_t *__restrict__ parted) {
    extern __shared__ uint32_t lds_ps[];
    uint32_t *cnt = lds_ps;                 // [nparts] -> local cursor
    uint32_t *lbase = lds_ps + nparts;      // [nparts] first staging slot of the partition
    uint32_t *stage = lds_ps + 2 * nparts;  // [CHUNK] entries sorted by partition
    uint16_t *spid = reinterpret_cast<uint16_t *>(stage + CHUNK);  // [CHUNK] partition of each slot
    __shared__ uint32_t scan_tmp[1];
    const uint32_t chunk = blockIdx.x, k = blockIdx.y, nchunks = gridDim.x, t = threadIdx.x, T = blockDim.x;
    const uint32_t *goff = blockhist + ((size_t)k * nchunks + chunk) * nparts;  // prefix of (chunk, p) inside partition p
    const uint32_t *pbase = part_base + (size_t)k * (nparts + 1);
    // The chunk's population of every partition is already known: k_part_hist counted it and k_part_colscan turned the
    // counts into prefixes over the chunks, so it is the difference to the next chunk's prefix (the last chunk: to the
    // partition's population) - no counting pass of LDS atomics over the entries (measured against that pass in round 4:
    // profiles/r04_scatter_count_ab.log).
    {
        const bool last = chunk + 1 == nchunks;
        for (uint32_t p = t; p < nparts; p += T) cnt[p] = (last ? pbase[p + 1] - pbase[p] : goff[nparts + p]) - goff[p];
    }
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < n ? lo + chunk_len : n;
    const D *d = digits + (size_t)k * n;
    const uint32_t fmask = (1u << fbits) - 1u;
    static_assert(CHUNK % 1024 == 0, "a whole number of entries per thread");
    constexpr int PER = CHUNK / 1024;  // entries per thread, in registers while the cursors are prepared
    uint32_t ent[PER];
    uint32_t pid[PER];
    // A full chunk whose codes start on a 16-byte boundary is read as vectors of four codes per lane (round 6: three 8- /
    // 16-byte loads per thread instead of twelve 2- / 4-byte ones; the lanes of a wave still cover one contiguous block);
    // the last chunk of a window and unaligned windows (n not a multiple of 8) take the code-by-code loads.
    static_assert(PER % 4 == 0, "whole vectors of four codes per thread");
    const bool vec = hi - lo == CHUNK && T == 1024 && (reinterpret_cast<uintptr_t>(d + lo) & 15u) == 0;
    if (vec) {
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const size_t i0 = lo + ((size_t)q * T + t) * 4;
            uint32_t c4[4];
            if constexpr (sizeof(D) == 2) {
                const uint2 w = *reinterpret_cast<const uint2 *>(d + i0);
                c4[0] = w.x & 0xffffu, c4[1] = w.x >> 16, c4[2] = w.y & 0xffffu, c4[3] = w.y >> 16;
            } else {
                const uint4 w = *reinterpret_cast<const uint4 *>(d + i0);
                c4[0] = w.x, c4[1] = w.y, c4[2] = w.z, c4[3] = w.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * q + u;
                pid[j] = 0xFFFFFFFFu;
                if (c4[u]) {
                    const uint32_t b = code_bucket(c4[u]);
                    pid[j] = b >> fbits;
                    ent[j] = ((b & fmask) << lidx) | ((uint32_t)(i0 + u) << 1) | (c4[u] & 1u);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const size_t i = lo + (size_t)j * T + t;
            uint32_t code = i < hi ? d[i] : 0u;
            pid[j] = 0xFFFFFFFFu;
            if (code) {
                const uint32_t b = code_bucket(code);
                pid[j] = b >> fbits;
                ent[j] = ((b & fmask) << lidx) | ((uint32_t)i << 1) | (code & 1u);
            }
        }
    }
    __syncthreads();
    // exclusive scan of cnt[] -> lbase[]; cnt[] becomes the local cursor
    {
        const uint32_t tot = wave0_exclusive_scan(cnt, nparts, 0u, lbase, cnt);
        if (t == 0) scan_tmp[0] = tot;
    }
    __syncthreads();
    const uint32_t total = scan_tmp[0];
    const bool runny = wave_runny(pid[0], pid[0] != 0xFFFFFFFFu);  // one sample decides for the thread's PER entries
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool act = pid[j] != 0xFFFFFFFFu;
        const uint32_t slot = lds_count<true>(cnt, pid[j], act, runny);
        if (act) {
            stage[slot] = ent[j];
            spid[slot] = (uint16_t)pid[j];
        }
    }
    // lbase[p] becomes the displacement of partition p: staging slot -> address in `parted` (the placement above does not
    // read lbase), so that the write-out below looks nothing up in global memory - it had two dependent loads per entry
    for (uint32_t p = t; p < nparts; p += T) lbase[p] = pbase[p] + goff[p] - lbase[p];
    __syncthreads();
    uint32_t *out = parted + (size_t)k * n;
    for (uint32_t slot = t; slot < total; slot += T) out[lbase[spid[slot]] + slot] = stage[slot];
}

// grid = (nparts, nwin), block = 1024. Counting sort of one partition by fine bucket; also emits starts[] of its buckets.
// Partitions of more than stage_cap references are the heavy kernels' (below).
// Round 6: the partition is read ONCE - its references stay in registers (PER per thread, stage_cap <= 1024 PER) between the
// counting and the placing pass; rounds 3-5 read it from global memory for both (profiles/r06_front_end_ab.log).
template <int PER>
static __global__ void __launch_bounds__(1024) k_fine_sort(const uint32_t *__restrict__ parted, size_t n, uint32_t nbuckets,
                                                          uint32_t fbits, uint32_t lidx,
                                                          const uint32_t *__restrict__ part_base,
                                                          uint32_t *__restrict__ sorted, uint32_t *__restrict__ starts,
                                                          uint32_t stage_cap) {
    extern __shared__ uint32_t lds_f[];  // 2^fbits counters, then stage_cap staging slots
    const uint32_t p = blockIdx.x, k = blockIdx.y, nparts = gridDim.x, t = threadIdx.x, T = blockDim.x;
    const uint32_t nf = 1u << fbits;
    const uint32_t *pb = part_base + (size_t)k * (nparts + 1);
    const uint32_t lo = pb[p], hi = pb[p + 1];
    const uint32_t *in = parted + (size_t)k * n;
    uint32_t *out = sorted + (size_t)k * n;
    uint32_t *st = starts + (size_t)k * (nbuckets + 1);
    if (p == nparts - 1 && t == 0) st[nbuckets] = hi;
    const uint32_t pop = hi - lo;
    if (pop > stage_cap) return;  // oversized: k_heavy_hist / k_heavy_scan / k_heavy_place
    uint32_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {  // PER loads in flight per thread; the lanes of a wave read 256 contiguous bytes each
        const uint32_t e = lo + (uint32_t)j * T + t;
        v[j] = e < hi ? in[e] : 0u;
    }
    for (uint32_t f = t; f < nf; f += T) lds_f[f] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j += 4) {
        if (lo + (uint32_t)j * T >= hi) break;  // uniform: nothing left for any thread
        const bool runny = wave_runny(v[j] >> lidx, lo + (uint32_t)j * T + t < hi);
#pragma unroll
        for (int u = 0; u < 4 && j + u < PER; ++u)
            lds_count<false>(lds_f, v[j + u] >> lidx, lo + (uint32_t)(j + u) * T + t < hi, runny);
    }
    __syncthreads();
    // exclusive scan of the nf counters -> write cursors (in place) and starts[]
    wave0_exclusive_scan(lds_f, nf, lo, lds_f, st + (size_t)p * nf);
    __syncthreads();
    const uint32_t pmask = (1u << lidx) - 1u;
    // place the references inside LDS and write the sorted run out contiguously (scattered 4-byte global stores are
    // limited to well under one lane per cycle per CU; this path has none)
    uint32_t *stage = lds_f + nf;
#pragma unroll
    for (int j = 0; j < PER; j += 4) {
        if (lo + (uint32_t)j * T >= hi) break;
        const bool runny = wave_runny(v[j] >> lidx, lo + (uint32_t)j * T + t < hi);
#pragma unroll
        for (int u = 0; u < 4 && j + u < PER; ++u) {
            const bool act = lo + (uint32_t)(j + u) * T + t < hi;
            const uint32_t pos = lds_count<true>(lds_f, v[j + u] >> lidx, act, runny);
            if (act) stage[pos - lo] = v[j + u] & pmask;
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < pop; i += T) out[lo + i] = stage[i];
}

// ---- the sort of the oversized partitions (see HeavyPart above). subhist: [nwin][scap rows][2^fbits] counters.
// hcount[2k] = listed partitions of window k, hcount[2k + 1] = their sub-runs (= rows of the window's table), hcount[2 nwin] = the
// sub-runs of all windows (zeroed by k_part_colscan, summed by k_part_rowscan).
// Work items of k_heavy_hist / k_heavy_place = the rows of all windows, numbered window by window, taken round-robin by the
// workgroups of a 1-D grid (a window alone may hold all of them: the narrow top window of uniform scalars).
constexpr uint32_t HEAVY_MAX_WINDOWS = 256;
struct HeavyItem {
    uint32_t k, row, s0, s1;  // window, row of the window's table, the sub-run's references [s0, s1) of parted[k]
};
// Call from all threads of a 1024-thread workgroup. s_pref: [HEAVY_MAX_WINDOWS + 1] LDS words, s_hp: one LDS HeavyPart.
__device__ __forceinline__ uint32_t heavy_prefix(const uint32_t *hcount, uint32_t nw, uint32_t *s_pref) {
    const uint32_t t = threadIdx.x;
    if (t < nw) s_pref[t] = hcount[2 * t + 1];
    __syncthreads();
    const uint32_t tot = wave0_exclusive_scan(s_pref, nw, 0u, s_pref);
    if (t == 0) s_pref[nw] = tot;
    __syncthreads();
    return s_pref[nw];
}
__device__ __forceinline__ HeavyItem heavy_item(uint32_t g, uint32_t nw, const uint32_t *s_pref, HeavyPart *s_hp,
                                                const uint32_t *hcount, const HeavyPart *hparts, uint32_t hcap,
                                                const uint32_t *part_base, uint32_t nparts) {
    uint32_t lo = 0, hi = nw;  // largest k with s_pref[k] <= g
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pref[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t k = lo, row = g - s_pref[k], np = hcount[2 * k];
    for (uint32_t j = threadIdx.x; j < np; j += blockDim.x) {
        const HeavyPart hp = hparts[(size_t)k * hcap + j];
        if (row - hp.first < hp.nsub) *s_hp = hp;  // exactly one partition owns the row
    }
    __syncthreads();
    const HeavyPart hp = *s_hp;
    const uint32_t *pb = part_base + (size_t)k * (nparts + 1);
    const uint32_t plo = pb[hp.p], phi = pb[hp.p + 1];
    const uint32_t s0 = plo + (row - hp.first) * HEAVY_SUB;
    return HeavyItem{k, row, s0, s0 + HEAVY_SUB < phi ? s0 + HEAVY_SUB : phi};
}

// k_heavy_hist: grid = G, block = 1024, dynamic LDS = 4 << fbits. Leaves the fine-bucket counts of every sub-run in its row.
static __global__ void __launch_bounds__(1024) k_heavy_hist(const uint32_t *__restrict__ parted, size_t n, uint32_t nw, uint32_t nparts,
                                                           uint32_t fbits, uint32_t lidx,
                                                           const uint32_t *__restrict__ part_base,
                                                           const HeavyPart *__restrict__ hparts,
                                                           const uint32_t *__restrict__ hcount, uint32_t hcap, uint32_t scap,
                                                           uint32_t *__restrict__ subhist) {
    extern __shared__ uint32_t lds_h[];
    __shared__ uint32_t s_pref[HEAVY_MAX_WINDOWS + 1];
    __shared__ HeavyPart s_hp;
    const uint32_t t = threadIdx.x, T = blockDim.x;
    if (hcount[2 * nw] == 0) return;  // no oversized partition in this launch (the usual case): one load
    const uint32_t total = heavy_prefix(hcount, nw, s_pref);
    const uint32_t nf = 1u << fbits;
    constexpr uint32_t PER = HEAVY_SUB / 1024;
    for (uint32_t g = blockIdx.x; g < total; g += gridDim.x) {
        const HeavyItem it = heavy_item(g, nw, s_pref, &s_hp, hcount, hparts, hcap, part_base, nparts);
        const uint32_t *in = parted + (size_t)it.k * n;
        uint32_t v[PER];
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) {
            const uint32_t e = it.s0 + i * T + t;
            v[i] = e < it.s1 ? in[e] : 0xFFFFFFFFu;
        }
        for (uint32_t f = t; f < nf; f += T) lds_h[f] = 0;
        __syncthreads();
        const bool runny = wave_runny(v[0] >> lidx, it.s0 + t < it.s1);
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) lds_count<false>(lds_h, v[i] >> lidx, it.s0 + i * T + t < it.s1, runny);
        __syncthreads();
        uint32_t *row = subhist + ((size_t)it.k * scap + it.row) * nf;
        for (uint32_t f = t; f < nf; f += T) row[f] = lds_h[f];
        __syncthreads();
    }
}

// k_heavy_scan: grid = (G, nwin), block = 1024, dynamic LDS = 4 << fbits. One workgroup per listed partition: column sums
// of its rows -> exclusive scan over the fine buckets (starts[], from the partition's base) -> every row rewritten with
// the write cursors of its sub-run (bucket start + counts of the sub-runs before it). Thread (f, s) walks segment s of
// column f; adjacent threads take adjacent columns.
static __global__ void __launch_bounds__(1024) k_heavy_scan(uint32_t nparts, uint32_t nbuckets, uint32_t fbits,
                                                           const uint32_t *__restrict__ part_base,
                                                           const HeavyPart *__restrict__ hparts,
                                                           const uint32_t *__restrict__ hcount, uint32_t hcap, uint32_t scap,
                                                           uint32_t *__restrict__ subhist, uint32_t *__restrict__ starts) {
    extern __shared__ uint32_t lds_tot[];
    __shared__ uint32_t seg[1024];
    const uint32_t k = blockIdx.y, t = threadIdx.x;
    const uint32_t np = hcount[2 * k];
    const uint32_t nf = 1u << fbits;
    const uint32_t FC = nf < 1024u ? nf : 1024u, SEGS = 1024u / FC;
    const uint32_t col = t % FC, s = t / FC;
    const uint32_t *pb = part_base + (size_t)k * (nparts + 1);
    uint32_t *st = starts + (size_t)k * (nbuckets + 1);
    for (uint32_t j = blockIdx.x; j < np; j += gridDim.x) {
        const HeavyPart hp = hparts[(size_t)k * hcap + j];
        uint32_t *rows = subhist + ((size_t)k * scap + hp.first) * nf;
        const uint32_t per = (hp.nsub + SEGS - 1) / SEGS;
        const uint32_t a = s * per < hp.nsub ? s * per : hp.nsub, b = a + per < hp.nsub ? a + per : hp.nsub;
        for (uint32_t f0 = 0; f0 < nf; f0 += FC) {
            const uint32_t f = f0 + col;
            uint32_t mine = 0;
            for (uint32_t sub = a; sub < b; ++sub) mine += rows[(size_t)sub * nf + f];
            seg[t] = mine;
            __syncthreads();
            if (s == 0) {
                uint32_t tot = 0;
                for (uint32_t q = 0; q < SEGS; ++q) tot += seg[q * FC + col];
                lds_tot[f] = tot;
            }
            __syncthreads();  // (nf <= 1024: one pass, seg[] stays valid for the rewrite below)
        }
        wave0_exclusive_scan(lds_tot, nf, pb[hp.p], lds_tot, st + (size_t)hp.p * nf);
        __syncthreads();
        for (uint32_t f0 = 0; f0 < nf; f0 += FC) {
            const uint32_t f = f0 + col;
            uint32_t run = lds_tot[f];
            for (uint32_t q = 0; q < s; ++q) run += seg[q * FC + col];  // SEGS > 1 only when nf <= 1024
            for (uint32_t sub = a; sub < b; ++sub) {
                const uint32_t v = rows[(size_t)sub * nf + f];
                rows[(size_t)sub * nf + f] = run;
                run += v;
            }
        }
        __syncthreads();
    }
}

// k_heavy_place: grid and LDS as k_heavy_hist. The sub-run's cursors come from its row; a reference goes straight to its
// slot of `sorted` (a run of lanes that share a bucket holds consecutive slots: coalescing stores).
static __global__ void __launch_bounds__(1024) k_heavy_place(const uint32_t *__restrict__ parted, size_t n, uint32_t nw, uint32_t nparts,
                                                            uint32_t fbits, uint32_t lidx,
                                                            const uint32_t *__restrict__ part_base,
                                                            const HeavyPart *__restrict__ hparts,
                                                            const uint32_t *__restrict__ hcount, uint32_t hcap, uint32_t scap,
                                                            const uint32_t *__restrict__ subhist, uint32_t *__restrict__ sorted) {
    extern __shared__ uint32_t lds_h[];
    __shared__ uint32_t s_pref[HEAVY_MAX_WINDOWS + 1];
    __shared__ HeavyPart s_hp;
    const uint32_t t = threadIdx.x, T = blockDim.x;
    if (hcount[2 * nw] == 0) return;
    const uint32_t total = heavy_prefix(hcount, nw, s_pref);
    const uint32_t nf = 1u << fbits;
    const uint32_t pmask = (1u << lidx) - 1u;
    constexpr uint32_t PER = HEAVY_SUB / 1024;
    for (uint32_t g = blockIdx.x; g < total; g += gridDim.x) {
        const HeavyItem it = heavy_item(g, nw, s_pref, &s_hp, hcount, hparts, hcap, part_base, nparts);
        const uint32_t *in = parted + (size_t)it.k * n;
        uint32_t *out = sorted + (size_t)it.k * n;
        const uint32_t *row = subhist + ((size_t)it.k * scap + it.row) * nf;
        for (uint32_t f = t; f < nf; f += T) lds_h[f] = row[f];
        uint32_t v[PER];
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) {
            const uint32_t e = it.s0 + i * T + t;
            v[i] = e < it.s1 ? in[e] : 0xFFFFFFFFu;
        }
        __syncthreads();
        const bool runny = wave_runny(v[0] >> lidx, it.s0 + t < it.s1);
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) {
            const bool act = it.s0 + i * T + t < it.s1;
            const uint32_t pos = lds_count<true>(lds_h, v[i] >> lidx, act, runny);
            if (act) out[pos] = v[i] & pmask;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ bucket accumulation (the hot loop)
// The bases are first rewritten once per call (or once per registered SRS) into the lazy Montgomery
// domain (k_convert_points: 2 field multiplications per point, against 10 per mixed add per window), packed back into
// the same 2N words per point so the gather in the hot loop still moves 64 B per BN254 point.
template <class U>
struct UAffine {  // packed lazy-domain coordinates (values < 2q): same size as the Go affine point
    uint32_t x[LzTraits<U>::PACKED_WORDS], y[LzTraits<U>::PACKED_WORDS];
};

// The point at infinity, affine (0, 0) (g1.go:41-47), stays all-zero words through the rewrite (the Montgomery product of
// zero is zero, and no finite point has both coordinates divisible by q): the accumulation recognises it from the
// gathered record itself - one OR of the two low words for every entry, the full test only when that is zero - so the
// scalar pipeline of an unregistered call does not have to wait for the rewrite to learn which points to skip.
template <class U>
__device__ __forceinline__ bool uaffine_is_infinity(const UAffine<U> &p) {
    if ((p.x[0] | p.y[0]) != 0u) return false;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < (int)LzTraits<U>::PACKED_WORDS; ++i) any |= p.x[i] | p.y[i];
    return any == 0u;
}

template <class U>
__global__ void __launch_bounds__(256) k_convert_points(const void *__restrict__ points, size_t n, void *__restrict__ upoints,
                                                        uint8_t *__restrict__ skip) {
    using T = LzTraits<U>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<typename T::Sat> a = load_struct<Affine<typename T::Sat>>(points, i);
    if (skip != nullptr) skip[i] = a.is_infinity() ? 1 : 0;
    const U ux = T::template from_sat<true>(a.x), uy = T::template from_sat<true>(a.y);
    UAffine<U> u;
    T::pack(ux, u.x);
    T::pack(uy, u.y);
    store_struct(upoints, i, u);
}

// ------------------------------------------------------------------ entry-parallel segmented accumulation
// Load-balanced form of the accumulation: every thread takes SEG consecutive entries of a window's bucket-sorted
// reference list (not one bucket), so the work per lane is the same whatever the bucket-size distribution is
// (uniform scalars, the reference's "smallvalues"/"redundancy" benchmark distributions, multiexp_test.go:319-334, or
// all scalars equal).  A run of entries that lies completely inside the thread's range is a finished bucket and is
// stored directly; a run that continues into a neighbouring thread is stored as a partial (slot 0: run open to the
// left, slot 1: run open only to the right) and k_fixup_seg adds the chain of partials of each split bucket.
// Buckets without entries are never written: the reduction (and k_merge_buckets) recognise them from `starts`.
struct SegFlags {
    static constexpr uint32_t HAS_P0 = 1u;         // first run continues from the previous thread
    static constexpr uint32_t P0_OPEN_RIGHT = 2u;  // ... and also continues into the next thread
    static constexpr uint32_t HAS_P1 = 4u;         // last run starts in this thread and continues into the next
};

// waves per SIMD the accumulation kernel is compiled for: 3 for 9-limb coordinates (160 VGPRs, no spills), 2 for 14-limb
// ones, 1 (all 512 registers) beyond and for Fp2
#ifndef GMSM_W9
#define GMSM_W9 3
#endif
template <class U> struct AccWaves { static constexpr int value = 1; };
template <class P> struct AccWaves<FpU<P>> { static constexpr int value = P::UL <= 9 ? GMSM_W9 : (P::UL <= 14 ? 2 : 1); };
// Fp2: one wave per SIMD for both base fields. (BN254 G2 ran two waves on 256 registers through round 4; once the signed-limb
// addition needed 300, that form spilled 67-75 of them into a 212-byte scratch frame inside the loop: one wave on the full
// register file is 12 % faster - 2^20 accumulate 4.12 -> 3.62 ms, 2^22 15.3 -> 13.0, profiles/r05_g2_waves_ab.log.)
template <class P> struct AccWaves<Fp2U<P>> { static constexpr int value = 1; };

// The accumulation loop inlines its field products for every element type. (Tried for the two largest - Fp2 over 14
// limbs, 90 KB of loop body, and 28 limbs, 130 KB, both beyond the 64 KB instruction cache: calling one shared copy of
// the product instead is SLOWER, BW6-761 2^20 43.7 against 32.9 ms, BLS12-381 G2 2^22 55.1 against 47.0 ms - the calls
// spill 0.5-1 KB per lane. What those kernels were missing is instruction-level parallelism, see GMSM_MUL_NACC.)
// TAB (window tables of registered bases, Group::precompute_tables): the launch sees ONE window whose entry e = w * tab_m + i
// stands for 2^(c w) P_i, stored at slot w * tab_stride + i of `upoints` (tab_stride = registered points, tab_m = points
// of this call); a separate instantiation, so the loop of the ordinary path is untouched.
template <bool TAB>
__device__ __forceinline__ size_t point_slot(uint32_t index, uint32_t tab_m, uint32_t tab_stride) {
    if constexpr (TAB) {
        const uint32_t w = index / tab_m;
        return (size_t)w * tab_stride + (index - w * tab_m);
    } else {
        return index;
    }
}

template <class U, bool TAB = false>
__global__ void __launch_bounds__(256, AccWaves<U>::value) k_accumulate_seg(const void *__restrict__ upoints, size_t n, uint32_t nbuckets,
                                                           uint32_t seg, const uint32_t *__restrict__ starts,
                                                           const uint32_t *__restrict__ sorted, void *__restrict__ buckets,
                                                           void *__restrict__ partials, uint32_t *__restrict__ pflags,
                                                           uint32_t *__restrict__ pbucket, uint32_t threads_per_win,
                                                           uint32_t tab_m = 0, uint32_t tab_stride = 0) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    const uint32_t *st = starts + (size_t)k * (nbuckets + 1);
    const uint32_t total = st[nbuckets];
    const size_t tg = (size_t)k * threads_per_win + t;
    const uint64_t e0_64 = (uint64_t)t * seg;
    if (t >= threads_per_win) return;
    if (e0_64 >= total) {
        pflags[tg] = 0;
        return;
    }
    const uint32_t e0 = (uint32_t)e0_64;
    const uint32_t e1 = (total - e0 > seg) ? e0 + seg : total;
    // bucket containing entry e0: largest b with st[b] <= e0 (then st[b+1] > e0 because empty buckets have equal starts)
    uint32_t lo = 0, hi = nbuckets;  // invariant: st[lo] <= e0 < st[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (st[mid] <= e0) lo = mid; else hi = mid;
    }
    // Software pipeline: nothing the current addition needs is loaded in the iteration that uses it.
    //   entry e+2      (the address of the NEXT iteration's gather) is requested now;
    //   point of e+1   is gathered while the point of e is being added (64 B per lane; at 2^24 the 1 GiB table is HBM);
    //   starts[b+2]    (the end of the NEXT bucket) is requested when the current bucket ends.
    // Measured ablations (profiles/r02_accumulate_ablation.md): with the operands in registers the same additions take
    // 1.08-1.12 ms per 2^24 (tools/ubench_madd.hip), this kernel 1.35 ms at 2^20; removing the gather, the bucket flushes
    // and all boundary handling together only gets it to 1.28 ms. At 2^24 the gather (HBM instead of Infinity Cache)
    // costs 12 %; requesting the entries one iteration earlier, or gathering whole 64-byte lines with four cooperating
    // lanes through an LDS tile (9-13 % slower: spills), does not recover it.
    uint32_t b = lo;
    uint32_t bend = st[b + 1];
    uint32_t bend2 = st[b + 2 <= nbuckets ? b + 2 : nbuckets];
    bool open_left = st[b] < e0;
    uint32_t flags = 0;
    const uint32_t *ent = sorted + (size_t)k * n;
    using T = LzTraits<U>;
    XYZZL<U> acc;
    bool inf = true;
    uint32_t v = ent[e0];
    uint32_t vn = e0 + 1 < e1 ? ent[e0 + 1] : 0u;
    UAffine<U> p = load_struct<UAffine<U>>(upoints, point_slot<TAB>(v >> 1, tab_m, tab_stride));
    for (uint32_t e = e0; e < e1; ++e) {
        if (e == bend) {  // the current run is complete on the right
            lz_acc_finish<true>(acc, inf);  // a raw record: its readers finish it (lz_rec_fresh)
            if (open_left) {
                lazy_store<U>(partials, tg * 2 + 0, acc, inf);
                flags |= SegFlags::HAS_P0;
                open_left = false;
            } else {
                lazy_store<U>(buckets, (size_t)k * nbuckets + b, acc, inf);
            }
            inf = true;
            ++b;
            bend = bend2;
            // empty buckets (rare under uniform scalars: the next bucket's end was prefetched, further ones are not). A few
            // are stepped over; a long gap is bisected - when two crowded buckets of a window lie far apart (all scalars
            // equal under GLV: one bucket per half, up to 2^(c-1) apart) the ONE thread whose segment holds the boundary
            // walked the gap with a dependent load per bucket while the whole launch waited for it: 2.7-5.5 ms instead of
            // 1.2 at 2^20 (profiles/r06_glv_all_equal.log)
            for (int step = 0; step < 4 && bend == e; ++step) {
                ++b;
                bend = st[b + 1];
            }
            if (bend == e) {  // the bucket of entry e: largest b' with st[b'] <= e (st[b + 1] == e holds, st[nbuckets] = total > e)
                uint32_t glo = b + 1, ghi = nbuckets;
                while (ghi - glo > 1) {
                    const uint32_t mid = (glo + ghi) >> 1;
                    if (st[mid] <= e) glo = mid; else ghi = mid;
                }
                b = glo;
                bend = st[b + 1];
            }
            bend2 = st[b + 2 <= nbuckets ? b + 2 : nbuckets];
        }
        const uint32_t vc = v;
        const UAffine<U> pc = p;
        const uint32_t vnn = e + 2 < e1 ? ent[e + 2] : 0u;
        if (e + 1 < e1) p = load_struct<UAffine<U>>(upoints, point_slot<TAB>(vn >> 1, tab_m, tab_stride));
        v = vn;
        vn = vnn;
        if (!uaffine_is_infinity<U>(pc))  // affine (0, 0) contributes nothing (g1.go:825)
            lz_madd_acc<true>(acc, inf, T::unpack(pc.x), T::unpack(pc.y), (vc & 1u) != 0);
    }
    {
        const bool open_right = bend > e1;
        lz_acc_finish<true>(acc, inf);  // a raw record: its readers finish it (lz_rec_fresh)
        if (open_left) {
            lazy_store<U>(partials, tg * 2 + 0, acc, inf);
            flags |= SegFlags::HAS_P0 | (open_right ? SegFlags::P0_OPEN_RIGHT : 0u);
        } else if (open_right) {
            lazy_store<U>(partials, tg * 2 + 1, acc, inf);
            flags |= SegFlags::HAS_P1;
            pbucket[tg] = b;
        } else {
            lazy_store<U>(buckets, (size_t)k * nbuckets + b, acc, inf);
        }
    }
    pflags[tg] = flags;
}

// Chain fixup. A split bucket is a chain  P1[t0], P0[t0+1], ..., P0[t1]  of partial sums of consecutive accumulation
// threads; its length follows from starts[] alone: the bucket's entries [lo, hi) lie in the threads lo / seg .. (hi - 1) /
// seg.  k_fixup_seg: the thread that owns the chain head adds up to MAXWALK followers (random scalars: 1-3). A longer
// chain (a narrow top window - BW6-761's has 9 bits, 256 buckets of n/256 entries -, repeated scalars, every scalar
// equal) goes to a list as PIECES of at most LONG_PIECE links, and k_fixup_long gives every piece a workgroup of 64 lane
// quads: strided sums, then a tree - LONG_PIECE/64 + 6 quad steps; the workgroup that finishes a chain's last piece adds the
// pieces' sums the same way. (Round 1 closed long chains with two hierarchical passes that were serial inside their
// spans. Rounds 2-4 gave a whole chain to ONE workgroup, m/64 + 6 steps: fine for BW6-761's 9-link chains, 0.9 ms for the
// 13 K-link chain every window gets at 2^20 when all scalars are equal, 2.7 ms at 2^24 - profiles/r05_distributions_before.log.)
struct LongChain {
    uint32_t window, head;    // window index inside the launch, head thread (the chain's destination is pbucket[head])
    uint32_t piece, npieces;  // this item: links [piece * LONG_PIECE, ...) of the chain; items of a chain are consecutive
};
constexpr uint32_t FIXUP_MAXWALK = 8;  // upper limit of the `maxwalk` argument of k_fixup_seg
constexpr uint32_t LONG_PIECE = 256;   // links per item of the long-chain list

// Appends the pieces of one long chain of m links (m > 1) to the list; called by the one thread that found it.
__device__ __forceinline__ void long_chain_append(uint32_t *long_count, LongChain *long_list, uint32_t *piece_done,
                                                  uint32_t window, uint32_t head, uint32_t m) {
    const uint32_t np = (m + LONG_PIECE - 1) / LONG_PIECE;
    const uint32_t slot = atomicAdd(long_count, np);
    for (uint32_t i = 0; i < np; ++i) long_list[slot + i] = LongChain{window, head, i, np};
    piece_done[slot] = 0;  // pieces finished so far (k_fixup_long: the last one to finish adds the pieces up)
}

template <class A>
__global__ void __launch_bounds__(256) k_fixup_seg(uint32_t nbuckets, const void *__restrict__ partials,
                                                   const uint32_t *__restrict__ pflags, const uint32_t *__restrict__ pbucket,
                                                   uint32_t threads_per_win, void *__restrict__ buckets,
                                                   uint32_t *__restrict__ long_count, LongChain *__restrict__ long_list,
                                                   uint32_t *__restrict__ piece_done,
                                                   uint32_t maxwalk /* followers a head adds itself; longer chains -> list */,
                                                   const uint32_t *__restrict__ starts, uint32_t seg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (t >= threads_per_win) return;
    const size_t base = (size_t)k * threads_per_win;
    // The usual chain is P1[t] + P0[t+1]: fetch all of it (and the destination) before looking at any flag, so that the
    // dependent memory round trips of the straightforward walk overlap.
    const uint32_t f0 = pflags[base + t];
    const bool has_next = t + 1 < threads_per_win;
    typename A::Elem acc = A::load(partials, (base + t) * 2 + 1);  // raw: made fresh below, by the chain heads only
    typename A::Elem q = A::load(partials, (base + (has_next ? t + 1 : t)) * 2 + 0);
    const uint32_t dest = pbucket[base + t];
    if (!(f0 & SegFlags::HAS_P1)) return;
    // followers of the chain: the accumulation threads t + 1 .. (end of the bucket - 1) / seg
    const uint32_t followers = (starts[(size_t)k * (nbuckets + 1) + dest + 1] - 1u) / seg - t;
    if (followers > maxwalk) {  // long: handed over untouched
        long_chain_append(long_count, long_list, piece_done, k, t, followers + 1u);
        return;
    }
    A::fresh(acc);
    for (uint32_t u = 1; u <= followers; ++u) {
        if (u != 1) q = A::load(partials, (base + t + u) * 2 + 0);
        A::fresh(q);
        A::add(acc, q);
    }
    A::store(buckets, (size_t)k * nbuckets + dest, acc);
}

// The same fix-up with one thread per BUCKET, for a shared bucket set (window tables): there a bucket holds nwin times
// the entries (BN254 G1, 2^20 points, c = 17: 240 against a thread's 80), so nearly every bucket is a chain of 3-4 partial
// sums and k_fixup_seg's heads - one lane in three - walk them with the other lanes of their wave idle (0.100 ms).
// Chains longer than maxwalk go to the same list as before.
template <class A>
__global__ void __launch_bounds__(256) k_fixup_bucket(uint32_t nbuckets, const uint32_t *__restrict__ starts, uint32_t seg,
                                                      const void *__restrict__ partials, uint32_t threads_per_win,
                                                      void *__restrict__ buckets, uint32_t *__restrict__ long_count,
                                                      LongChain *__restrict__ long_list, uint32_t *__restrict__ piece_done,
                                                      uint32_t maxwalk) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (b >= nbuckets) return;
    const uint32_t *st = starts + (size_t)k * (nbuckets + 1);
    const uint32_t lo = st[b], hi = st[b + 1];
    if (hi <= lo) return;
    const uint32_t t0 = lo / seg, t1 = (hi - 1) / seg;
    if (t0 == t1) return;  // the bucket lies inside one thread's range: stored by the accumulation itself
    const size_t base = (size_t)k * threads_per_win;
    if (t1 - t0 > maxwalk) {
        long_chain_append(long_count, long_list, piece_done, k, t0, t1 - t0 + 1u);
        return;
    }
    typename A::Elem acc = A::load_fresh(partials, (base + t0) * 2 + 1);
    for (uint32_t u = t0 + 1; u <= t1; ++u) {
        const typename A::Elem q = A::load_fresh(partials, (base + u) * 2 + 0);
        A::add(acc, q);
    }
    A::store(buckets, (size_t)k * nbuckets + b, acc);
}

// grid = any (grid-stride over the list's items), block = 256 = 64 quads, dynamic LDS = 128 * sizeof(QRec<U>).
// Additions on lane quads with the operands in LDS (gmsm_quad.h): quad j first adds up the links j, j + 64, ... of the
// item's piece (each fetched into the quad's staging record, one coordinate per lane), then a tree over the 64 quads. A
// chain of one piece is stored to its bucket; otherwise the sum goes to piece_sums[item] and the workgroup whose
// increment of the chain's counter completes it (release / acquire at device scope around the counter) adds the pieces'
// sums the same way. (Round 2 ran this with one-lane additions out of line: 2.9 KB of scratch per lane for the 28-limb
// field, 0.77 ms for the 256 nine-link chains of BW6-761's top window.)
template <class U>
__device__ __forceinline__ void quad_sum_links(QRec<U> *acc, QRec<U> *stage, uint32_t m, uint32_t tid, const void *src,
                                               size_t first_index, size_t stride, size_t head_index, bool fresh_records) {
    // sums the m records src[idx(i)], idx(i) = first_index + i * stride (idx(0) = head_index when that is not ~0), into acc[0]
    const uint32_t j = tid >> 2, lane = tid & 63u;
    if ((tid & 3u) == 0) acc[j].inf = 1u;
    for (uint32_t i0 = 0; i0 < m; i0 += 64) {
        const uint32_t i = i0 + j;
        const size_t idx = (i == 0 && head_index != ~(size_t)0) ? head_index : first_index + (size_t)i * stride;
        if (fresh_records) quad_rec_load<U, true>(&stage[j], src, idx, i < m, lane);
        else quad_rec_load<U, false>(&stage[j], src, idx, i < m, lane);
        __syncthreads();  // the lanes of a quad exchange coordinates through the record: stores before loads
        const QAddOps<U> o = quad_add_load<U>(&acc[j], &stage[j], lane);  // both records belong to this quad
        quad_add_store<U, true>(&acc[j], o, i < m, lane);
    }
    __syncthreads();
    uint32_t active = 64;
    while (active / 2 >= m && active > 1) active >>= 1;  // smallest power of two >= min(m, 64)
#pragma nounroll
    for (uint32_t d = active >> 1; d >= 1; d >>= 1) {
        const bool act = j < d;
        const QAddOps<U> o = quad_add_load<U>(&acc[j], &acc[act ? j + d : j], lane);
        __syncthreads();
        quad_add_store<U, true>(&acc[j], o, act, lane);
        __syncthreads();
    }
}

template <class U>
__global__ void __launch_bounds__(256) k_fixup_long(uint32_t nbuckets, const void *__restrict__ partials,
                                                    const uint32_t *__restrict__ pbucket, uint32_t threads_per_win,
                                                    void *__restrict__ buckets, const uint32_t *__restrict__ long_count,
                                                    const LongChain *__restrict__ long_list, uint32_t *__restrict__ piece_done,
                                                    void *__restrict__ piece_sums, const uint32_t *__restrict__ starts,
                                                    uint32_t seg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *acc = reinterpret_cast<QRec<U> *>(lds_raw), *stage = acc + 64;
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, j = tid >> 2, lane = tid & 63u, count = *long_count;
    for (uint32_t c = blockIdx.x; c < count; c += gridDim.x) {
        const LongChain lc = long_list[c];
        const size_t base = (size_t)lc.window * threads_per_win;
        const uint32_t dest = pbucket[base + lc.head];
        // chain length m (head included): the bucket's entries end in thread (hi - 1) / seg
        const uint32_t m = (starts[(size_t)lc.window * (nbuckets + 1) + dest + 1] - 1u) / seg - lc.head + 1u;
        const uint32_t l0 = lc.piece * LONG_PIECE;
        const uint32_t len = m - l0 < LONG_PIECE ? m - l0 : LONG_PIECE;
        // link i of the chain is P1[head] for i = 0 and P0[head + i] otherwise
        quad_sum_links<U>(acc, stage, len, tid, partials, (base + lc.head + l0) * 2 + 0, 2,
                          l0 == 0 ? (base + lc.head) * 2 + 1 : ~(size_t)0, true);
        const size_t bucket_index = (size_t)lc.window * nbuckets + dest;
        if (lc.npieces == 1) {
            if (j == 0) quad_rec_store<U>(buckets, bucket_index, &acc[0], lane);
            __syncthreads();
            continue;
        }
        const uint32_t c0 = c - lc.piece;  // the chain's first item: its counter and the base of its pieces' sums
        if (j == 0) quad_rec_store<U>(piece_sums, c, &acc[0], lane);
        __syncthreads();
        if (tid == 0) {
            __threadfence();  // the piece's sum is visible device-wide before the counter says so
            const uint32_t done = __hip_atomic_fetch_add(&piece_done[c0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = done + 1u == lc.npieces ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            quad_sum_links<U>(acc, stage, lc.npieces, tid, piece_sums, (size_t)c0, 1, ~(size_t)0, false);
            if (j == 0) quad_rec_store<U>(buckets, bucket_index, &acc[0], lane);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ bucket reduction
// Weighted sum  sum_k (k+1) B_k  of a window, as a two-level segmented running sum:
//   level 1: each thread of k_reduce_serial owns L consecutive buckets (running sum, multiexp_jacobian.go:44-52 restricted
//   to its segment) and leaves S_t = sum of the segment, W_t = sum_{k in seg t} (k-lo_t+1) B_k; k_combine_q (gmsm_quad.h)
//   combines N consecutive (S_t, W_t) with a suffix scan and two trees on lane quads;
//   level 2 (k_reduce2_q): one block per window combines the level-1 block results the same way.
// Identity:  sum_{k in [base, base+T*L)} (k-base+1) B_k = sum_t W_t + L * sum_{t>=1} Suf_t,  Suf_t = sum_{t'>=t} S_t'.
// (Rounds 1-2 fused the serial part and a one-lane combine into one kernel for the 9-limb field - k_reduce1, 36 dependent
// one-lane additions at c = 16. Split, with the combine on quads, measured 0.29 against 0.34 ms at 2^16 and 0.44 against
// 0.47 ms at 2^24; equal at 2^20.)

// The serial part of level 1: thread g of a window owns the L = 2^log2L
// consecutive buckets [g*L, (g+1)*L) and runs the reference's running sum over them (multiexp_jacobian.go:44-52), leaving
// S_g = sum B_j and W_g = sum (j+1) B_j in pre[(k*T + g)*2 + {0,1}], T = ceil(nbuckets / L). Fused with the combine (rounds 1-2) the
// loop keeps four extended-Jacobian values plus the addition's temporaries alive - for a 28-limb field or Fp2 over 14
// limbs that is over 600 registers and the kernel ran out of a 3-4 KB per lane scratch frame (BW6-761: 9.1 ms for a 4.8 ms
// multiplier bill). Here only `run`, `tot` and the loaded bucket are live, the addition is inlined, and the launch is
// not tied to the LDS-limited workgroups of the combine step, so it spreads over every SIMD.
template <class A>
__global__ void __launch_bounds__(256, 1) k_reduce_serial(const void *__restrict__ buckets, uint32_t nbuckets, uint32_t log2L,
                                                          uint32_t T, const uint32_t *__restrict__ starts,
                                                          void *__restrict__ pre) {
    using E = typename A::Elem;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (g >= T) return;
    const uint32_t L = 1u << log2L, lo = g * L;
    const uint32_t *st = starts ? starts + (size_t)k * (nbuckets + 1) : nullptr;  // null: every record is stored
    E run = A::infinity(), tot = A::infinity();
    // The bucket of the NEXT step is requested before the two additions of this one (the kernel runs one wave per SIMD, so
    // nothing else would hide the round trip of a load issued where its value is needed). Measured: no difference at
    // 2^16..2^20 (profiles/r04_reduce_prefetch.log) - the L round trips per thread are not what the 0.12 ms are made of.
    auto present = [&](uint32_t b) { return b < nbuckets && (st == nullptr || st[b + 1] > st[b]); };  // empty buckets were never written
    bool have_next = present(lo + L - 1);
    E next = A::infinity();
    if (have_next) next = A::load(buckets, (size_t)k * nbuckets + lo + L - 1);
#pragma nounroll
    for (uint32_t j = L; j-- > 0;) {
        const bool have = have_next;
        E B = next;
        have_next = j > 0 && present(lo + j - 1);
        if (have_next) next = A::load(buckets, (size_t)k * nbuckets + lo + j - 1);
        if (have) {
            A::fresh(B);  // a record the accumulation wrote
            A::add(run, B);
        }
        A::add(tot, run);
    }
    A::store(pre, ((size_t)k * T + g) * 2 + 0, run);
    A::store(pre, ((size_t)k * T + g) * 2 + 1, tot);
}

// The serial part of level 1 on lane quads, for the element types whose one-lane addition is so long (64 us per step for
// the 28-limb field, out of a 512-register frame with spills) that a window's buckets do not fill the chip with one-lane
// threads: quad g owns the L = 2^log2L buckets [g L, (g+1) L) and keeps `run` and `tot` in LDS records; a bucket is
// fetched into the quad's staging record, one coordinate per lane. A quad addition takes about a third of the one-lane
// time, so with the same number of lanes busy the chain is 4/3.3 as long per bucket - but a window of 2^13 buckets now
// occupies four times as many lanes, and the number of (S, W) pairs left for the combine is a quarter.
// Every record belongs to one quad: no workgroup barriers, only the ordering of a wave's own LDS accesses.
// grid = (ceil(T / QPB), nwin), block = 4 QPB, dynamic LDS = 3 QPB * sizeof(QRec<U>); output as k_reduce_serial.
#ifndef GMSM_SERIAL_Q_WAVES
#define GMSM_SERIAL_Q_WAVES 1  // waves per SIMD the register allocation leaves room for (A/B: tools/build_ab.sh)
#endif
template <class U, int QPB /* quads per workgroup: 3 QPB records of LDS, 4 QPB threads */>
__global__ void __launch_bounds__(4 * QPB, GMSM_SERIAL_Q_WAVES) k_reduce_serial_q(const void *__restrict__ buckets, uint32_t nbuckets, uint32_t log2L,
                                                             uint32_t T, const uint32_t *__restrict__ starts,
                                                             void *__restrict__ pre) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *run = reinterpret_cast<QRec<U> *>(lds_raw), *tot = run + QPB, *stage = run + 2 * QPB;
    const uint32_t tid = threadIdx.x, j = tid >> 2, lane = tid & 63u, k = blockIdx.y;
    const uint32_t g = blockIdx.x * QPB + j;
    const uint32_t L = 1u << log2L, lo = g * L;
    const uint32_t *st = starts ? starts + (size_t)k * (nbuckets + 1) : nullptr;  // null: every record is stored
    if ((tid & 3u) == 0) {
        run[j].inf = 1u;
        tot[j].inf = 1u;
    }
    quad_lds_fence();
#pragma nounroll
    for (uint32_t jj = L; jj-- > 0;) {
        const uint32_t b = lo + jj;
        const bool present = g < T && b < nbuckets && (st == nullptr || st[b + 1] > st[b]);  // empty buckets were never written
        quad_rec_load<U, true>(&stage[j], buckets, (size_t)k * nbuckets + (present ? b : 0), present, lane);
        quad_lds_fence();
        {
            const QAddOps<U> o = quad_add_load<U>(&run[j], &stage[j], lane);
            quad_add_store<U, true>(&run[j], o, present, lane);
        }
        quad_lds_fence();
        {
            const QAddOps<U> o = quad_add_load<U>(&tot[j], &run[j], lane);
            quad_add_store<U, true>(&tot[j], o, g < T, lane);
        }
        quad_lds_fence();
    }
    if (g < T) {
        quad_rec_store<U>(pre, ((size_t)k * T + g) * 2 + 0, &run[j], lane);
        quad_rec_store<U>(pre, ((size_t)k * T + g) * 2 + 1, &tot[j], lane);
    }
}

// Multi-range host calls (Group::window_sums_from_host): every point range leaves its bucket sums, this kernel adds them to
// the call's running buckets, and the reduction runs once. One thread per (window, bucket); `init`: the first range
// initialises the running buckets - every record is stored from here on, infinity as zz = 0.
// The reference's split of a MultiExp in two halves adds the halves' RESULTS (multiexp.go:98-140); adding bucket by bucket
// gives the same group element and pays the reduction once.
template <class A>
__global__ void __launch_bounds__(256) k_merge_buckets(void *__restrict__ carry, const void *__restrict__ buckets,
                                                       const uint32_t *__restrict__ starts, uint32_t nbuckets, int init) {
    using E = typename A::Elem;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (b >= nbuckets) return;
    const uint32_t *st = starts + (size_t)k * (nbuckets + 1);
    const bool present = st[b + 1] > st[b];
    const size_t idx = (size_t)k * nbuckets + b;
    if (init) {
        E v = A::infinity();
        if (present) v = A::load_fresh(buckets, idx);
        A::store(carry, idx, v);
    } else if (present) {
        E c = A::load(carry, idx);
        const E v = A::load_fresh(buckets, idx);
        A::add(c, v);
        A::store(carry, idx, c);
    }
}

// ------------------------------------------------------------------ window tables of registered bases
// Group::precompute_tables: slab w of the table holds 2^(c w) P_i in the packed lazy-domain layout of the bases. One step
// slab w-1 -> slab w: c doublings per point (this kernel, lazy XYZZ record out), k_batch_normalize, k_convert_points.
// A point whose multiple is the identity (zz = 0: an even-order point, which no subgroup base is) leaves an infinity
// record; the host then finds the slab's infinity flags different from the bases' and gives the tables up.
template <class U, bool INL>
__global__ void __launch_bounds__(256) k_table_double(const void *__restrict__ slab, const uint8_t *__restrict__ skip, size_t n,
                                                      uint32_t c, void *__restrict__ recs) {
    using T = LzTraits<U>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZL<U> p;
    bool inf = skip[i] != 0;
    if (!inf) {
        const UAffine<U> a = load_struct<UAffine<U>>(slab, i);
        p.x = T::unpack(a.x);
        p.y = T::unpack(a.y);
        p.zz = p.zzz = lz_one((const U *)nullptr);
#pragma nounroll
        for (uint32_t l = 0; l < c; ++l) p = lz_pdbl<INL>(p);  // double_u / double_g return the class they take
        inf = T::template to_sat<INL>(p.zz).is_zero();
    }
    lazy_store<U>(recs, i, p, inf);
}

// flag[0] |= 1 when the infinity flags of a table slab differ from the bases'
static __global__ void __launch_bounds__(256) k_skip_mismatch(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, size_t n,
                                                       uint32_t *__restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (a[i] != 0) != (b[i] != 0)) atomicOr(flag, 1u);
}

// rows [first, first + count) of the window totals <- infinity (a shared-bucket launch has one total, in row 0)
template <class Final>
__global__ void k_fill_infinity(void *__restrict__ totals, uint32_t first, uint32_t count) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < count) policy_store<Final>(totals, first + r, Final::infinity());
}

}  // namespace gmsm
