// Device check of the LDS counter helpers of the sort passes (gmsm_kernels.h: run_heads / lds_count_runs / lds_count):
// for several key patterns and activity masks every active lane must receive a distinct slot of its key's counter, the
// slots of a key must be 0 .. count-1, and the counters must end at the keys' populations - with and without the run
// aggregation, from full and from partially active waves (lanes dropping out from the top, as in the kernels' loops).
//   hipcc --offload-arch=gfx950 -O2 -I gnark-crypto_amd/csrc -o tests/hip/count_runs_check tests/hip/count_runs_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gmsm_kernels.h"

using namespace gmsm;

constexpr uint32_t NKEYS = 64, T = 256, ROUNDS = 8;

template <bool RET>
__global__ void k_check(const uint32_t *keys, const uint8_t *act, uint32_t n_active_threads, int mode, uint32_t *slots, uint32_t *final_cnt) {
    __shared__ uint32_t cnt[NKEYS];
    const uint32_t t = threadIdx.x;
    if (t < NKEYS) cnt[t] = 0;
    __syncthreads();
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        if (t < n_active_threads) {  // lanes drop out from the top
            const uint32_t key = keys[r * T + t];
            const bool a = act[r * T + t] != 0;
            const bool runny = mode == 0 ? false : mode == 1 ? true : wave_runny(key, a);
            const uint32_t s = lds_count<RET>(cnt, key, a, runny);
            slots[r * T + t] = a ? s : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    if (t < NKEYS) final_cnt[t] = cnt[t];
}

int main() {
    uint32_t *d_keys, *d_slots, *d_cnt;
    uint8_t *d_act;
    const size_t N = (size_t)ROUNDS * T;
    if (hipMalloc(&d_keys, N * 4) || hipMalloc(&d_slots, N * 4) || hipMalloc(&d_cnt, NKEYS * 4) || hipMalloc(&d_act, N)) return 2;
    std::vector<uint32_t> keys(N), slots(N), cnt(NKEYS);
    std::vector<uint8_t> act(N);
    int bad = 0, cases = 0;
    srand(12345);
    for (int pattern = 0; pattern < 6; ++pattern)
        for (int actpat = 0; actpat < 3; ++actpat)
            for (uint32_t nthr : {256u, 200u, 65u, 1u})
                for (int mode = 0; mode < 3; ++mode)
                    for (int ret = 0; ret < 2; ++ret) {
                        for (size_t i = 0; i < N; ++i) {
                            switch (pattern) {
                                case 0: keys[i] = rand() % NKEYS; break;
                                case 1: keys[i] = 7; break;
                                case 2: keys[i] = (uint32_t)(i / 25) % NKEYS; break;
                                case 3: keys[i] = rand() % 32 ? 7 : rand() % NKEYS; break;
                                case 4: keys[i] = i % 5 ? rand() % NKEYS : 7; break;
                                default: keys[i] = (uint32_t)(i / 3) % NKEYS; break;
                            }
                            act[i] = actpat == 0 ? 1 : actpat == 1 ? (rand() % 4 != 0) : (i % 7 != 3);
                        }
                        (void)hipMemcpy(d_keys, keys.data(), N * 4, hipMemcpyHostToDevice);
                        (void)hipMemcpy(d_act, act.data(), N, hipMemcpyHostToDevice);
                        (void)hipMemset(d_slots, 0xff, N * 4);
                        if (ret) hipLaunchKernelGGL((k_check<true>), dim3(1), dim3(T), 0, 0, d_keys, d_act, nthr, mode, d_slots, d_cnt);
                        else hipLaunchKernelGGL((k_check<false>), dim3(1), dim3(T), 0, 0, d_keys, d_act, nthr, mode, d_slots, d_cnt);
                        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
                        (void)hipMemcpy(slots.data(), d_slots, N * 4, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(cnt.data(), d_cnt, NKEYS * 4, hipMemcpyDeviceToHost);
                        std::vector<uint32_t> pop(NKEYS, 0);
                        std::vector<std::vector<uint8_t>> seen(NKEYS);
                        for (uint32_t r = 0; r < ROUNDS; ++r)
                            for (uint32_t t = 0; t < nthr; ++t)
                                if (act[r * T + t]) ++pop[keys[r * T + t]];
                        bool ok = true;
                        for (uint32_t k = 0; k < NKEYS; ++k) {
                            if (cnt[k] != pop[k]) ok = false;
                            seen[k].assign(pop[k], 0);
                        }
                        if (ret && ok)
                            for (uint32_t r = 0; r < ROUNDS && ok; ++r)
                                for (uint32_t t = 0; t < nthr && ok; ++t)
                                    if (act[r * T + t]) {
                                        const uint32_t k = keys[r * T + t], s = slots[r * T + t];
                                        if (s >= pop[k] || seen[k][s]) ok = false;
                                        else seen[k][s] = 1;
                                    }
                        ++cases;
                        if (!ok) {
                            ++bad;
                            if (bad < 10) printf("MISMATCH pattern %d actpat %d threads %u mode %d ret %d\n", pattern, actpat, nthr, mode, ret);
                        }
                    }
    printf("count_runs_check: %d cases, %d bad\n", cases, bad);
    return bad ? 1 : 0;
}
