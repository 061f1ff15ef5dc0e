// Cost of LDS counter updates under key collisions inside a wave, and of the wave-aggregated forms that avoid them
// (lds_count_agg, gmsm_kernels.h). One workgroup of 1024 threads per CU-slot bumps 512 LDS counters ITER times per lane
// with keys of a given pattern; variants: 0 plain ds_add_rtn, 1 one lean peel round (readfirstlane + ballot), 2 two lean
// rounds, 3 DPP run detector + peel loop (the first form tried). Prints ns per wave-level update.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ldsagg tools/ubench_ldsagg.hip && tools/ubench_ldsagg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mask_rank(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
template <int ROUNDS, uint32_t MINC>
__device__ __forceinline__ uint32_t agg_lean(uint32_t *cnt, uint32_t key, bool active) {
    uint32_t res = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t rem = __ballot(active);
        if (rem == 0ull) break;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);  // undefined for inactive first lane: see below
        const int first = __ffsll((unsigned long long)rem) - 1;
        const uint32_t k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
        (void)k0;
        const bool mine = active && key == k1;
        const uint64_t m = __ballot(mine);
        const uint32_t c = (uint32_t)__popcll((unsigned long long)m);
        if (c < MINC) break;
        uint32_t base = 0;
        if ((int)__lane_id() == first) base = atomicAdd(&cnt[k1], c);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
        if (mine) {
            res = base + mask_rank(m);
            active = false;
        }
    }
    if (active) res = atomicAdd(&cnt[key], 1u);
    return res;
}
__device__ __forceinline__ uint32_t agg_dpp(uint32_t *cnt, uint32_t key, bool active) {
    uint32_t res = 0;
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x111, 0xf, 0xf, false);
    const uint64_t runs = __ballot(active && prev == key);
    if ((uint32_t)__popcll((unsigned long long)runs) >= 16) {
        uint64_t rem = __ballot(active);
#pragma nounroll
        for (uint32_t round = 0; round < 8 && rem != 0ull; ++round) {
            const int first = __ffsll((unsigned long long)rem) - 1;
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
            const bool mine = active && key == k0;
            const uint64_t m = __ballot(mine);
            const uint32_t c = (uint32_t)__popcll((unsigned long long)m);
            if (c < 4) break;
            uint32_t base = 0;
            if ((int)__lane_id() == first) base = atomicAdd(&cnt[k0], c);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
            if (mine) res = base + mask_rank(m);
            if (mine) active = false;
            rem &= ~m;
        }
    }
    if (active) res = atomicAdd(&cnt[key], 1u);
    return res;
}

// Variant 4: aggregation of RUNS of equal keys in adjacent lanes (heads = lanes whose key differs from the lane before,
// row_shr:1, so every 16-lane row starts a run): the head adds the run's length, the others take base + rank through one
// ds_bpermute. Gated per call by the number of heads (in the kernels: per batch of entries, from a sample).
__device__ __forceinline__ uint64_t run_heads(uint32_t keyx) {
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~keyx, (int)keyx, 0x111, 0xf, 0xf, false);
    return __ballot(prev != keyx);
}
__device__ __forceinline__ uint32_t agg_runs(uint32_t *cnt, uint32_t key, bool active, uint64_t H) {
    const uint32_t lane = __lane_id();
    const uint64_t le = ~0ull >> (63u - lane);            // lanes <= this one
    const uint32_t head = 63u - (uint32_t)__clzll((long long)(H & le));
    const uint64_t above = H & ~le;
    const uint32_t next = above ? (uint32_t)__ffsll((long long)above) - 1u : 64u;
    uint32_t base = 0;
    if (lane == head && active) base = atomicAdd(&cnt[key], next - head);
    base = (uint32_t)__shfl((int)base, (int)head, 64);
    return base + (lane - head);
}
__device__ __forceinline__ uint32_t agg_gated(uint32_t *cnt, uint32_t key, bool active) {
    const uint32_t keyx = active ? key : 0xFFFFFFFFu;
    const uint64_t H = run_heads(keyx);
    if (__popcll((unsigned long long)H) <= 16) return agg_runs(cnt, key, active, H);
    return active ? atomicAdd(&cnt[key], 1u) : 0u;
}

// pattern: 0 uniform over 512 keys, 1 all lanes equal, 2 runs of 25 lanes, 3 96 % one key, 4 20 % one key (every 5th lane)
template <int PAT>
__device__ __forceinline__ uint32_t make_key(uint32_t tid, uint32_t it) {
    uint32_t h = (tid * 0x9E3779B9u) ^ (it * 0x85EBCA6Bu);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    if (PAT == 0) return h & 511u;
    if (PAT == 1) return (it + (tid >> 30)) & 511u;
    if (PAT == 2) return ((tid + it * 1024u) / 25u * 0x9E3779B9u >> 23) & 511u;
    if (PAT == 3) return (h & 31u) == 0 ? (h >> 8) & 511u : 7u;
    return (tid % 5u) == 0 ? 7u : (h & 511u);
}

template <int VAR, int PAT>
__global__ void __launch_bounds__(1024) k_bench(uint32_t iters, uint32_t *sink) {
    __shared__ uint32_t cnt[512];
    if (threadIdx.x < 512) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t key = make_key<PAT>(threadIdx.x, it + blockIdx.x * 7u);
        if (VAR == 0) acc += atomicAdd(&cnt[key], 1u);
        else if (VAR == 1) acc += agg_lean<1, 24>(cnt, key, true);
        else if (VAR == 2) acc += agg_lean<2, 16>(cnt, key, true);
        else if (VAR == 3) acc += agg_dpp(cnt, key, true);
        else acc += agg_gated(cnt, key, true);
    }
    __syncthreads();
    if (acc == 0xdeadbeefu) sink[0] = acc + cnt[threadIdx.x & 511];
}

template <int VAR, int PAT>
static void run(const char *name, uint32_t *sink) {
    const uint32_t iters = 2000, blocks = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_bench<VAR, PAT>), dim3(blocks), dim3(1024), 0, 0, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-level updates: blocks * 16 waves * iters; 2 workgroups per CU on 256 CUs at once
    printf("%-28s variant %d: %8.3f ms  %6.2f ns per wave update per CU\n", name, VAR, best, best * 1e6 / ((double)blocks / 256 * 16 * iters));
}

int main() {
    uint32_t *sink;
    if (hipMalloc(&sink, 4096) != hipSuccess) return 1;
#define ALLV(P, N) run<0, P>(N, sink); run<1, P>(N, sink); run<2, P>(N, sink); run<3, P>(N, sink); run<4, P>(N, sink);
    ALLV(0, "uniform over 512 keys")
    ALLV(1, "all lanes equal")
    ALLV(2, "runs of 25 lanes")
    ALLV(3, "96 % one key")
    ALLV(4, "every 5th lane one key")
    return 0;
}
