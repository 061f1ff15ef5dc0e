// The split-bucket fix-up (k_fixup_seg, gmsm_kernels.h) on lane quads, for the element types whose one-lane addition
// does not fit the register file: BW6-761's 28-limb field and the Fp2 groups run k_fixup_seg out of 512 registers with
// 17-40 of them spilled, one wave per SIMD, 64 us per addition (BW6-761 2^20: 322 K two-link chains in five rounds of
// workgroups, 0.75 ms = 3.8 % of the call). A quad addition (gmsm_quad.h: operands in LDS, one product per lane and level)
// takes about a third of the time on a third of the registers, and a workgroup of 64 quads closes 64 chains at once.
//
// Same contract as k_fixup_seg: the quad of accumulation thread t closes the chain  P1[t], P0[t+1], ..., P0[t+m]  of the
// bucket that thread t left open to the right, when it has at most `maxwalk` followers; longer chains go to the list that
// k_fixup_long consumes (as pieces, long_chain_append). Every record a quad touches is its own (acc[j], stage[j]): no workgroup barriers, only the
// ordering of a wave's own LDS accesses.
// grid = (ceil(threads_per_win / 64), nwin), block = 256, dynamic LDS = 128 * sizeof(QRec<U>).
#pragma once
#include "gmsm_kernels.h"

namespace gmsm {

template <class U>
__global__ void __launch_bounds__(256) k_fixup_seg_q(uint32_t nbuckets, const void *__restrict__ partials,
                                                     const uint32_t *__restrict__ pflags, const uint32_t *__restrict__ pbucket,
                                                     uint32_t threads_per_win, void *__restrict__ buckets,
                                                     uint32_t *__restrict__ long_count, LongChain *__restrict__ long_list,
                                                     uint32_t *__restrict__ piece_done, uint32_t maxwalk,
                                                     const uint32_t *__restrict__ starts, uint32_t seg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *acc = reinterpret_cast<QRec<U> *>(lds_raw), *stage = acc + 64;
    const uint32_t tid = threadIdx.x, j = tid >> 2, lane = tid & 63u, k = blockIdx.y;
    const uint32_t t = blockIdx.x * 64u + j;
    const size_t base = (size_t)k * threads_per_win;
    const uint32_t f0 = t < threads_per_win ? pflags[base + t] : 0u;
    bool head = (f0 & SegFlags::HAS_P1) != 0;
    // followers of the chain, from starts[] (the four lanes of a quad read the same words: uniform in the quad): the
    // bucket's entries end in accumulation thread (hi - 1) / seg
    uint32_t len = 0, dest = 0;
    if (head) {
        dest = pbucket[base + t];
        len = (starts[(size_t)k * (nbuckets + 1) + dest + 1] - 1u) / seg - t;
        if (len > maxwalk) {  // a long chain: handed over untouched
            if ((tid & 3u) == 0) long_chain_append(long_count, long_list, piece_done, k, t, len + 1u);
            head = false;
        }
    }
    quad_rec_load<U, true>(&acc[j], partials, (base + t) * 2 + 1, head, lane);
    quad_lds_fence();
#pragma nounroll
    for (uint32_t u = 1; u <= maxwalk; ++u) {
        const bool act = head && u <= len;
        if (__ballot(act) == 0ull) break;  // no quad of this wave has a link left (the lengths only run out, never resume)
        quad_rec_load<U, true>(&stage[j], partials, (base + t + (act ? u : 0u)) * 2 + 0, act, lane);
        quad_lds_fence();
        const QAddOps<U> o = quad_add_load<U>(&acc[j], &stage[j], lane);
        quad_add_store<U, true>(&acc[j], o, act, lane);
        quad_lds_fence();
    }
    if (head) quad_rec_store<U>(buckets, (size_t)k * nbuckets + dest, &acc[j], lane);
}

}  // namespace gmsm
