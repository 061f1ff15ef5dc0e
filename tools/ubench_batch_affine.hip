// Batch-affine bucket accumulation on gfx950, prototyped and measured (round 4; SURVEY.md §8 a6:
// processChunkG1BatchAffine ecc/bn254/multiexp_affine.go:24-231, batchAddG1Affine ecc/bn254/g1.go:1122-1182).
//
// The reference adds a batch of independent affine pairs (P_k, Q_k) with ONE shared inversion (Montgomery's trick):
//   forward   dx_k = x2_k - x1_k,  prefix_k = dx_0 ... dx_(k-1)                          1 product per pair
//   invert    inv = (dx_0 ... dx_(B-1))^-1                                                one inversion per batch
//   backward  1/dx_k = inv * prefix_k, inv *= dx_k, lambda = (y2 - y1)/dx_k,
//             x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1                         5 products per pair
// i.e. 6 products per addition against the 10 (9 reductions) of the extended-Jacobian mixed addition the pipeline's
// k_accumulate_seg runs. On a CPU core the batch is a few hundred pairs in L1. On this GPU the unit that shares an
// inversion is a LANE (an instruction costs a wave the same whether one lane or 64 need it, so sharing an inversion
// across lanes buys nothing: the wave pays the ~380 products of x^(q-2) either way), so a lane needs B pairs per inversion
// with B >= 64 for the inversion to cost ~1 product equivalent per pair x 6, and the state of B pairs (prefix products, and
// the points again in the backward pass) does not fit registers or LDS at the occupancy the gather needs.
//
// This program measures exactly that trade on BN254's base field with the pipeline's own lazy-limb arithmetic
// (gmsm_fieldu.h):
//   k_inverse        per-lane x^(q-2): the cost I of one inversion in products
//   k_tree_pass<B>   one pass of the pairwise tree the round-3 verdict sketched: every lane takes B pairs of a pair list
//                    (random indices into a 2^20-point table = the first level's gather; consecutive records = the
//                    later levels), prefix products in a global scratch laid out [k][limb][thread] (coalesced),
//                    per-lane inversion, backward pass, affine sums written as 72-byte lazy records
//   k_xyzz_chain     the same number of additions as extended-Jacobian chains with the same gather (what the shipped
//                    kernel does per entry, without its bucket bookkeeping)
// and checks every batch-affine sum against the extended-Jacobian result of the same pair (x3 ZZ == X3, y3 ZZZ == Y3:
// the two formulas are the same rational functions, on or off the curve).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_batch_affine.hip -o tools/ubench_batch_affine
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gnark-crypto_amd/csrc/gmsm_curveu.h"
using namespace gmsm;
using P = bn254_fp_params;
using U = FpU<P>;
constexpr int L = P::UL, W = P::UW;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- K q in normalised limbs (K = 16, 32), for the top-limb conditional subtraction that keeps coordinates below 18q
struct KQ { uint32_t l[L]; };
constexpr KQ kq_limbs(unsigned K) {
    KQ r{};
    uint64_t c = 0;
    for (int i = 0; i < L; ++i) {
        const uint64_t t = (uint64_t)P::UQ1[i] * K + c;
        r.l[i] = i < L - 1 ? (uint32_t)(t & ((1u << W) - 1u)) : (uint32_t)t;
        c = t >> W;
    }
    return r;
}
constexpr KQ KQ16 = kq_limbs(16), KQ32 = kq_limbs(32);

template <int K>
__device__ __forceinline__ void cond_sub_kq(U &v) {  // v >= K q + 2^(W(L-1)) => v -= K q (top-limb test, borrows pre-distributed)
    constexpr const KQ &T = K == 16 ? KQ16 : KQ32;
    const uint32_t m = v.l[L - 1] > T.l[L - 1] ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < L; ++i) v.l[i] += m & ((i < L - 1 ? (1u << W) : 0u) - (i > 0 ? 1u : 0u) - T.l[i]);
    fpu_carry(v);
}
__device__ __forceinline__ U reduce_lt18(U v) {  // any value < 66q -> < 16q + 2^(W(L-1)) < 18q
    cond_sub_kq<32>(v);
    cond_sub_kq<16>(v);
    return v;
}

// ---- point records
struct Packed64 { uint32_t w[16]; };                 // x | y, 8 saturated words each, value < 2^256 (the pipeline's `upoints`)
struct Lazy72 { uint32_t l[18]; };                   // x | y, 9 lazy limbs each, value < 18q (output of a tree level)
__device__ __forceinline__ void load_point(const Packed64 *t, uint32_t i, U &x, U &y) {
    const uint4 *s = reinterpret_cast<const uint4 *>(t + i);
    uint4 v[4] = {s[0], s[1], s[2], s[3]};
    const uint32_t *w = reinterpret_cast<const uint32_t *>(v);
    x = fpu_unpack<P>(w);
    y = fpu_unpack<P>(w + 8);
}
__device__ __forceinline__ void load_point(const Lazy72 *t, uint32_t i, U &x, U &y) {
    const uint2 *s = reinterpret_cast<const uint2 *>(t + i);  // 72 = 9 x 8 bytes
    uint2 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = s[k];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(v);
#pragma unroll
    for (int k = 0; k < L; ++k) x.l[k] = w[k], y.l[k] = w[L + k];
}
__device__ __forceinline__ void store_point(Lazy72 *t, size_t i, const U &x, const U &y) {
    uint2 v[9];
    uint32_t *w = reinterpret_cast<uint32_t *>(v);
#pragma unroll
    for (int k = 0; k < L; ++k) w[k] = x.l[k], w[L + k] = y.l[k];
    uint2 *d = reinterpret_cast<uint2 *>(t + i);
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = v[k];
}

// ---- x^(q-2), square-and-multiply over the constant exponent (uniform branches): 253 squarings + ~110 products
__device__ __forceinline__ U fpu_inverse(const U &a) {
    uint32_t e[P::N];
#pragma unroll
    for (int i = 0; i < P::N; ++i) e[i] = P::Q[i];
    e[0] -= 2u;  // q is odd and its low word is far from 0/1: no borrow
    U r = a;
    int top = 32 * P::N - 1;
    while (!((e[top >> 5] >> (top & 31)) & 1u)) --top;
#pragma nounroll
    for (int b = top - 1; b >= 0; --b) {
        r = fpu_sqr(r);
        if ((e[b >> 5] >> (b & 31)) & 1u) r = fpu_mul(r, a);
    }
    return r;
}

__global__ void __launch_bounds__(256) k_inverse(const uint32_t *in, uint32_t *out, int reps) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    U a;
#pragma unroll
    for (int i = 0; i < L; ++i) a.l[i] = (in[i] + tid * 0x9e3779b9u) & (i < L - 1 ? FpU<P>::MASK : 0xffffu);
    U r = a;
    for (int k = 0; k < reps; ++k) r = fpu_inverse(r);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) s ^= r.l[i];
    out[tid] = s;
}

// ---- one pass of the pairwise tree. Lane t owns pairs [t*B, (t+1)*B) of the list; GATHER: the pair's points are
// table[idx[2p]], table[idx[2p+1]] (first level), else consecutive records 2p, 2p+1 (later levels).
// prefix scratch: [k][limb][thread] uint32 (one coalesced dword per limb and lane).
template <class Rec, int B, bool GATHER>
__global__ void __launch_bounds__(256) k_tree_pass(const Rec *__restrict__ table, const uint32_t *__restrict__ idx, size_t npairs,
                                                   uint32_t *__restrict__ scratch, Lazy72 *__restrict__ out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t p0 = tid * B;
    if (p0 >= npairs) return;
    U run = fpu_one<P>();
    // forward: prefix products of the x differences
#pragma nounroll
    for (int k = 0; k < B; ++k) {
        const size_t p = p0 + k;
        const uint32_t i0 = GATHER ? idx[2 * p] : (uint32_t)(2 * p), i1 = GATHER ? idx[2 * p + 1] : (uint32_t)(2 * p + 1);
        U x1, y1, x2, y2;
        load_point(table, i0, x1, y1);
        load_point(table, i1, x2, y2);
        const U dx = fpu_sub<P, 32>(x2, x1);  // < 50q
#pragma unroll
        for (int i = 0; i < L; ++i) scratch[((size_t)k * L + i) * nthreads + tid] = run.l[i];
        run = fpu_mul(run, dx);               // 2 * 50 / 169 + 1 < 2
    }
    U inv = fpu_inverse(run);
    // backward
#pragma nounroll
    for (int k = B - 1; k >= 0; --k) {
        const size_t p = p0 + k;
        const uint32_t i0 = GATHER ? idx[2 * p] : (uint32_t)(2 * p), i1 = GATHER ? idx[2 * p + 1] : (uint32_t)(2 * p + 1);
        U x1, y1, x2, y2, pre;
        load_point(table, i0, x1, y1);
        load_point(table, i1, x2, y2);
#pragma unroll
        for (int i = 0; i < L; ++i) pre.l[i] = scratch[((size_t)k * L + i) * nthreads + tid];
        const U dx = fpu_sub<P, 32>(x2, x1);
        const U idx_inv = fpu_mul(inv, pre);                   // 1 / dx_k
        inv = fpu_mul(inv, dx);
        const U lam = fpu_mul(fpu_sub<P, 32>(y2, y1), idx_inv);  // < 2
        U x3 = fpu_sub<P, 32>(fpu_sub<P, 32>(fpu_sqr(lam), x1), x2);  // < 66q
        x3 = reduce_lt18(x3);
        U y3 = fpu_sub<P, 32>(fpu_mul(lam, fpu_sub<P, 32>(x1, x3)), y1);  // < 34q
        cond_sub_kq<16>(y3);
        store_point(out, p, x3, y3);
    }
}

// ---- the same additions as extended-Jacobian chains (one chain of `seg` gathered points per lane), the pipeline's way
template <class Rec, bool GATHER>
__global__ void __launch_bounds__(256, 3) k_xyzz_chain(const Rec *__restrict__ table, const uint32_t *__restrict__ idx, size_t nadds, int seg,
                                                        uint32_t *__restrict__ out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t e0 = tid * seg;
    if (e0 >= nadds) return;
    XYZZU<P> acc;
    bool inf = true;
    for (int k = 0; k < seg; ++k) {
        const size_t e = e0 + k;
        U x, y;
        load_point(table, GATHER ? idx[e] : (uint32_t)e, x, y);
        madd_u<P, true>(acc, inf, x, y, false);
    }
    uint32_t s = inf;
#pragma unroll
    for (int i = 0; i < L; ++i) s ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.zz.l[i] ^ acc.zzz.l[i];
    out[tid] = s;
}

// ---- check: out[p] == table[i0] + table[i1] through the extended-Jacobian mixed addition (as rational functions)
template <class Rec, bool GATHER>
__global__ void k_check(const Rec *table, const uint32_t *idx, size_t npairs, const Lazy72 *out, unsigned long long *bad) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const uint32_t i0 = GATHER ? idx[2 * p] : (uint32_t)(2 * p), i1 = GATHER ? idx[2 * p + 1] : (uint32_t)(2 * p + 1);
    U x1, y1, x2, y2, x3, y3;
    load_point(table, i0, x1, y1);
    load_point(table, i1, x2, y2);
    load_point(out, (uint32_t)p, x3, y3);
    // madd_u wants px < 2, py < 6: bring the operands down with a product by one (exact mod q)
    const U one = fpu_one<P>();
    XYZZU<P> acc;
    bool inf = true;
    madd_u<P, true>(acc, inf, fpu_mul(x1, one), fpu_mul(y1, one), false);
    madd_u<P, true>(acc, inf, fpu_mul(x2, one), fpu_mul(y2, one), false);
    const Fp<P> lx = fpu_to_sat<P>(fpu_mul(fpu_mul(x3, one), acc.zz)), rx = fpu_to_sat<P>(acc.x);
    const Fp<P> ly = fpu_to_sat<P>(fpu_mul(fpu_mul(y3, one), acc.zzz)), ry = fpu_to_sat<P>(acc.y);
    bool ok = !inf;
    for (int i = 0; i < P::N; ++i) ok = ok && lx.l[i] == rx.l[i] && ly.l[i] == ry.l[i];
    if (!ok) atomicAdd(bad, 1ull);
}

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    void start() { (void)hipEventRecord(a); }
    float stop() { (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

template <class Rec, int B, bool GATHER>
static int run_tree(const char *name, const Rec *table, const uint32_t *idx, size_t npairs, uint32_t *scratch, Lazy72 *out,
                    unsigned long long *d_bad, double inv_products) {
    const size_t threads = npairs / B;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    Timer t;
    k_tree_pass<Rec, B, GATHER><<<blocks, 256>>>(table, idx, npairs, scratch, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(d_bad, 0, 8));
    k_check<Rec, GATHER><<<(unsigned)((npairs + 255) / 256), 256>>>(table, idx, npairs, out, d_bad);
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    t.start();
    const int reps = 3;
    for (int r = 0; r < reps; ++r) k_tree_pass<Rec, B, GATHER><<<blocks, 256>>>(table, idx, npairs, scratch, out);
    const float ms = t.stop() / reps;
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, (const void *)k_tree_pass<Rec, B, GATHER>));
    const double per_add = 6.0 + inv_products / B;
    const double bytes = npairs * (2.0 * 2 * sizeof(Rec) + 2.0 * 36 + 72 + (GATHER ? 16 : 0));
    printf("%-34s B=%4d lanes=%8zu vgpr=%3d scratch=%4d  %8.3f ms  %6.2f G adds/s  %5.2f products/add -> %6.1f G products/s  %5.2f TB/s  mismatches=%llu\n",
           name, B, threads, fa.numRegs, (int)fa.localSizeBytes, ms, npairs / ms * 1e-6, per_add, npairs * per_add / ms * 1e-6, bytes / ms * 1e-9, bad);
    return 0;
}

int main(int argc, char **argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 20;        // table size (points): 2^20 = 64 MiB sits in the Infinity Cache, 2^24 does not
    const int logp = argc > 2 ? atoi(argv[2]) : 23;        // pairs per pass: 2^23 pairs = 2^24 entries, one window's worth at 2^24 points
    const size_t n = (size_t)1 << logn, npairs = (size_t)1 << logp;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; table 2^%d points, 2^%d pair-additions per pass\n", prop.name, prop.multiProcessorCount, logn, logp);

    // random field elements below 2^253 < q as "points": the addition formulas are rational identities, no curve needed
    std::vector<Packed64> h_tab(n);
    for (auto &r : h_tab)
        for (int i = 0; i < 16; ++i) r.w[i] = (uint32_t)rnd() & ((i & 7) == 7 ? 0x1fffffffu : 0xffffffffu);
    std::vector<uint32_t> h_idx(2 * npairs);
    for (auto &v : h_idx) v = (uint32_t)(rnd() >> 11) & (uint32_t)(n - 1);
    for (size_t p = 0; p < npairs; ++p)
        if (h_idx[2 * p] == h_idx[2 * p + 1]) h_idx[2 * p + 1] ^= 1u;  // equal x: the pipeline diverts such pairs to the doubling path
    Packed64 *d_tab;
    uint32_t *d_idx, *d_scr, *d_out32, *d_in;
    Lazy72 *d_out, *d_out2;
    unsigned long long *d_bad;
    CHECK(hipMalloc(&d_tab, n * sizeof(Packed64)));
    CHECK(hipMalloc(&d_idx, 2 * npairs * 4));
    CHECK(hipMalloc(&d_scr, npairs * L * 4));          // prefix products of one pass: 36 B per pair
    CHECK(hipMalloc(&d_out, npairs * sizeof(Lazy72)));
    CHECK(hipMalloc(&d_out2, npairs / 2 * sizeof(Lazy72)));
    CHECK(hipMalloc(&d_out32, (size_t)1 << 24));
    CHECK(hipMalloc(&d_in, 64));
    CHECK(hipMalloc(&d_bad, 8));
    CHECK(hipMemcpy(d_tab, h_tab.data(), n * sizeof(Packed64), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_idx, h_idx.data(), 2 * npairs * 4, hipMemcpyHostToDevice));
    uint32_t h_in[16];
    for (auto &v : h_in) v = (uint32_t)rnd();
    CHECK(hipMemcpy(d_in, h_in, 64, hipMemcpyHostToDevice));

    // ---- 1. the inversion: per-lane x^(q-2) against the bare product rate
    Timer t;
    const int blocks = prop.multiProcessorCount * 8;
    k_inverse<<<blocks, 256>>>(d_in, d_out32, 1);
    CHECK(hipDeviceSynchronize());
    t.start();
    k_inverse<<<blocks, 256>>>(d_in, d_out32, 4);
    const float ms_inv = t.stop() / 4;
    const double inversions = (double)blocks * 256;
    const double products_per_s = 174e9;  // bare lazy product on this chip (tools/ubench_fpmul.hip, profiles/peaks_r02.json)
    const double inv_products = ms_inv * 1e-3 / inversions * products_per_s;
    printf("per-lane inversion x^(q-2): %.3f ms for %.0f inversions = %.2f G inversions/s = %.0f product-times each (at %.0f G products/s)\n",
           ms_inv, inversions, inversions / ms_inv * 1e-6, inv_products, products_per_s * 1e-9);

    // ---- 2. the extended-Jacobian chains on the same gather (the shipped way, without bucket bookkeeping)
    for (int seg : {32, 64}) {
        const size_t nadds = 2 * npairs;
        const size_t threads = nadds / seg;
        k_xyzz_chain<Packed64, true><<<(unsigned)((threads + 255) / 256), 256>>>(d_tab, d_idx, nadds, seg, d_out32);
        CHECK(hipDeviceSynchronize());
        t.start();
        for (int r = 0; r < 3; ++r) k_xyzz_chain<Packed64, true><<<(unsigned)((threads + 255) / 256), 256>>>(d_tab, d_idx, nadds, seg, d_out32);
        const float ms = t.stop() / 3;
        printf("%-34s seg=%3d lanes=%8zu  %8.3f ms  %6.2f G adds/s  10 products/add -> %6.1f G products/s\n", "xyzz chains, gathered 64 B records", seg,
               threads, ms, nadds / ms * 1e-6, nadds * 10.0 / ms * 1e-6);
    }

    // ---- 3. tree passes: first level (gather from the packed table), later level (consecutive 72-byte records)
    if (run_tree<Packed64, 16, true>("tree level 1 (gather)", d_tab, d_idx, npairs, d_scr, d_out, d_bad, inv_products)) return 1;
    if (run_tree<Packed64, 64, true>("tree level 1 (gather)", d_tab, d_idx, npairs, d_scr, d_out, d_bad, inv_products)) return 1;
    if (run_tree<Packed64, 256, true>("tree level 1 (gather)", d_tab, d_idx, npairs, d_scr, d_out, d_bad, inv_products)) return 1;
    if (run_tree<Packed64, 1024, true>("tree level 1 (gather)", d_tab, d_idx, npairs, d_scr, d_out, d_bad, inv_products)) return 1;
    // level 2 input = level 1 output (npairs records -> npairs/2 pairs)
    k_tree_pass<Packed64, 64, true><<<(unsigned)((npairs / 64 + 255) / 256), 256>>>(d_tab, d_idx, npairs, d_scr, d_out);
    CHECK(hipDeviceSynchronize());
    if (run_tree<Lazy72, 16, false>("tree level 2 (consecutive records)", d_out, nullptr, npairs / 2, d_scr, d_out2, d_bad, inv_products)) return 1;
    if (run_tree<Lazy72, 64, false>("tree level 2 (consecutive records)", d_out, nullptr, npairs / 2, d_scr, d_out2, d_bad, inv_products)) return 1;
    if (run_tree<Lazy72, 256, false>("tree level 2 (consecutive records)", d_out, nullptr, npairs / 2, d_scr, d_out2, d_bad, inv_products)) return 1;
    return 0;
}
