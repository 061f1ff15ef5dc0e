// FETCH_SIZE calibration for the access pattern of k_accumulate_seg's base gather: every lane reads ONE 64-byte record
// (4 x 16-byte loads, like load_struct<UAffine>) at a pseudo-random index of a table that is far larger than the
// 256 MiB Infinity Cache, so every record is a compulsory 64-byte read from HBM.  Known bytes = lanes x 64.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
//   rocprofv3 --pmc FETCH_SIZE -d <dir> -o g --output-format csv -- tools/ubench_gather [log2_records] [log2_lanes]
// and compare FETCH_SIZE (KiB) of k_gather64 with the figure the program prints (profiles/r03_fetch_calibration.md).
// A second kernel (k_stream16) reads the same table as a coalesced 16-B-per-lane stream: the guide's factor-2 case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_gather64(const uint4 *__restrict__ table, unsigned log2_records, size_t lanes,
                                                  uint4 *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    // multiplication by an odd constant is a bijection of [0, 2^k): no record is read twice (lanes <= records)
    const size_t rec = (size_t)((i * 0x9E3779B97F4A7C15ull) & (((unsigned long long)1 << log2_records) - 1));
    const uint4 *p = table + rec * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint4 s = {a.x ^ b.x ^ c.x ^ d.x, a.y ^ b.y ^ c.y ^ d.y, a.z ^ b.z ^ c.z ^ d.z, a.w ^ b.w ^ c.w ^ d.w};
    if (s.x == 0x12345678u && s.y == 0x9abcdef0u) sink[i & 1023] = s;  // never true for the zero-filled table
}

__global__ void __launch_bounds__(256) k_stream16(const uint4 *__restrict__ table, size_t count, uint4 *__restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 s = {0, 0, 0, 0};
    for (; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = table[i];
        s.x ^= a.x; s.y ^= a.y; s.z ^= a.z; s.w ^= a.w;
    }
    if (s.x == 0x12345678u && s.y == 0x9abcdef0u) sink[threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const unsigned log2_records = argc > 1 ? atoi(argv[1]) : 26;  // 2^26 x 64 B = 4 GiB
    const unsigned log2_lanes = argc > 2 ? atoi(argv[2]) : 24;
    const size_t records = (size_t)1 << log2_records, lanes = (size_t)1 << log2_lanes;
    uint4 *table, *sink;
    if (hipMalloc(&table, records * 64) != hipSuccess || hipMalloc(&sink, 1024 * 16) != hipSuccess) return 1;
    (void)hipMemset(table, 0, records * 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather64, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, 0, table, log2_records, lanes, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather64: %zu lanes x 64 B = %.1f MiB known bytes, %.3f ms, %.0f GB/s\n", lanes, lanes * 64.0 / 1048576.0, ms,
               lanes * 64.0 / ms / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, table, records * 4, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_stream16: %.1f MiB known bytes, %.3f ms, %.0f GB/s\n", records * 64.0 / 1048576.0, ms, records * 64.0 / ms / 1e6);
    }
    return 0;
}
