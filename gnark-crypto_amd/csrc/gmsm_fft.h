// fr/fft on the device (second half): the transform that sits on the other side of the MSM in a
// prover pipeline (polynomial coefficients <-> evaluations; the MSM then commits to either).
//
// Replaces, with identical results (every output is a uniquely determined field element, compared limb for limb):
//   Domain / NewDomain            ecc/bn254/fr/fft/domain.go:24-110   (Generator: fr/generator.go:18-36, coset shift :56-62)
//   (*Domain).FFT / FFTInverse    ecc/bn254/fr/fft/fft.go:31-196      (DIF: natural in, bit-reversed out; DIT: the reverse;
//                                                                      OnCoset: scale by the powers of FrMultiplicativeGen)
//   difFFT / ditFFT               fft.go:198-330                      (radix-2 butterflies, fr.Butterfly = (a+b, a-b))
//   BitReverse                    ecc/bn254/fr/fft/bitreverse.go:20-45
// and the twins of BLS12-381 and BW6-761 (same templates, other scalar fields).
//
// Shape on the GPU: the reference recurses over halves with one goroutine per half; here the log2 n radix-2 stages run
// as a few passes over HBM, each pass doing up to 8-11 stages on LDS-resident tiles (k_fft_pass; 2^24: three passes
// instead of 24), with the twiddles w^t for t < n/2 resident per domain like the reference's precomputed tables. The
// butterflies run on lazy 29-bit limbs (gmsm_fft_lazy.h: one v_mad_u64_u32 per partial product, values reduced by a
// top-limb test between products, canonical again on every store), and the coset / 1/n scalings ride on the first
// load and the last store of the transform instead of being passes of their own.
#pragma once
#include <hip/hip_runtime.h>
#include "gmsm_context.h"
#include "gmsm_field.h"
#include "gmsm_fft_lazy.h"

namespace gmsm {

template <class FrP>
struct FftPowers {  // base^(2^b), b < 40: lets every thread form base^i with <= log2(i) products
    Fp<FrP> p[40];
};

template <class T>
__device__ __forceinline__ T fft_load(const T *a, size_t i) {
    static_assert(sizeof(T) % 16 == 0, "element size");
    T r;
    const uint4 *src = reinterpret_cast<const uint4 *>(a + i);
    uint4 *dst = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); ++k) dst[k] = src[k];
    return r;
}
template <class T>
__device__ __forceinline__ void fft_store(T *a, size_t i, const T &v) {
    uint4 *dst = reinterpret_cast<uint4 *>(a + i);
    const uint4 *src = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); ++k) dst[k] = src[k];
}

__device__ __forceinline__ size_t fft_bitrev(size_t i, unsigned log2n) {
    return log2n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log2n)) : 0;
}

// out[i] = scale * base^e, i < count, e = i (BuildExpTable, fft/domain.go; here one thread per entry) or, rev_bits != 0,
// e = i with its low rev_bits bits reversed: the twiddle table in the order the top-down passes read it.
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_pow_table(FftPowers<FrP> pw, Fp<FrP> scale, size_t count, Fp<FrP> *__restrict__ out,
                                                       unsigned rev_bits = 0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const size_t e = rev_bits ? fft_bitrev(i, rev_bits) : i;
    Fp<FrP> acc = scale;
#pragma nounroll
    for (int b = 0; b < 40; ++b)
        if ((e >> b) & 1) acc = fp_mul(acc, pw.p[b]);
    fft_store(out, i, acc);
}

// a[i] *= table[rev ? bitrev(i) : i] (coset scaling; the table may carry 1/n folded in) and a[i] *= c (CardinalityInv,
// fft.go:144-150), tables / constants in the lazy domain (2^DOMAIN_SHIFT * factor, gmsm_fft_lazy.h). Launched on their
// own only by the A/B paths and for n = 1: k_fft_pass_lz fuses them into its loads and stores.
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_scale_table_lz(Fp<FrP> *__restrict__ a, size_t n, unsigned log2n,
                                                            const Fp<FrP> *__restrict__ table, int rev) {
    using Z = FftLz<FrP>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t t = rev ? fft_bitrev(i, log2n) : i;
    fft_store(a, i, Z::store(Z::mul(Z::load(fft_load(a, i)), fft_load(table, t))));
}
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_scale_const_lz(Fp<FrP> *__restrict__ a, size_t n, Fp<FrP> c) {
    using Z = FftLz<FrP>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fft_store(a, i, Z::store(Z::mul(Z::load(fft_load(a, i)), c)));
}

// Several consecutive radix-2 stages in one pass over HBM. A stage pairs elements whose indices differ in one bit b. A
// pass takes the B stages of bits [bl, bl+B) and a workgroup owns a tile of 2^B x C elements - every value of those B
// bits x C consecutive values of the low bits (C x 32 bytes contiguous per row, so the strided passes still move whole
// 128-256-byte runs) - stages it in LDS, runs the B stages with a barrier in between and writes it back: one read and
// one write of the vector per pass instead of per stage (2^24: 3 passes instead of 24 trips through HBM).
//
// Round 4 - ONE butterfly for every transform. The kernel is bound by VALU issue (rocprofv3: SQ_ACTIVE_INST_VALU = 96 %
// of the kernel's cycles, ~400 instructions per butterfly of which 190 are the product, profiles/r04_fft_stats.md), so
// what counts is instructions per butterfly. The reference's DIF is Gentleman-Sande - (x, y) <- (x + y, (x - y) w) - whose
// sums double the bound of a value at every stage, so every stage needs a conditional subtraction; its DIT is
// Cooley-Tukey - t = y w, (x, y) <- (x + t, x - t) - where a bound grows by 2q per stage and a whole pass needs none
// (FftLz::dit_free). The forward order "natural in, bit-reversed out" has a Cooley-Tukey form too: bit b from the top
// down, the pair (i, i + 2^b) of block k = i >> (b + 1) takes the twiddle w^(bitrev_(log n - 1)(k)) - the SAME block
// index in every stage, so one table in bit-reversed order (FftDomain::twiddles_rev_lz) serves all of them, and the top
// passes, whose blocks are few, read a handful of entries where the Gentleman-Sande form streamed the whole table. Every
// output is the same field element whichever flow graph computes it, so parity is untouched (the A/B and the bit-exact
// suite: profiles/r04_fft_ab.log). BN254 2^24: DIF 2.59 -> 2.3x ms, DIT 2.58 -> 2.36.
//   TOPDOWN = true : stages from bit bl + B - 1 down (FFT/FFTInverse with decimation DIF), twz = bit-reversed table
//   TOPDOWN = false: stages from bit bl up (decimation DIT), twz = natural table, entry j << (log n - 1 - b)
// The stage whose twiddles are all one (the first of the transform: bit log n - 1 top-down, bit 0 bottom-up) runs without
// its product (dit_one). Values in the tile are below 40q (FftLz); they come back below 2q once per pass, on the store:
// the canonical element from the transform's last pass, any representative that fits the element from the others.
// Scalings of the transform fused into the pass that touches the vector first / last:
//   pre  != null : every element is multiplied by pre[pre_rev ? bitrev(i) : i] as it is loaded (coset FFT, fft.go:43-82)
//   post_mode 1/2: ... by post[i] / post[bitrev(i)] as it is stored (inverse coset FFT, fft.go:153-195)
//   post_mode 3  : ... by the constant post_c (CardinalityInv, fft.go:144-150)
// Tile of the contiguous low pass: 2^10 elements (36 KB of LDS for a 32-byte field: four workgroups per CU instead of the
// two that 2^11 allowed) and 512 threads per tile from 2^22 elements on (profiles/r03_fft_tiles.log).
#ifndef GMSM_FFT_LOWB
#define GMSM_FFT_LOWB 10
#endif
template <class FrP, bool TOPDOWN>
__global__ void __launch_bounds__(512) k_fft_pass_lz(Fp<FrP> *__restrict__ a, unsigned log2n, unsigned bl, unsigned B, unsigned log2C,
                                                     const Fp<FrP> *__restrict__ twz, const Fp<FrP> *__restrict__ pre, int pre_rev,
                                                     const Fp<FrP> *__restrict__ post, int post_mode, Fp<FrP> post_c, int final_pass) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using Z = FftLz<FrP>;
    using U = FpU<FrP>;
    U *tile = reinterpret_cast<U *>(lds_raw);  // [2^B][C]
    const unsigned C = 1u << log2C, T = blockDim.x, t = threadIdx.x;
    const size_t ntile_lo = ((size_t)1 << bl) >> log2C;
    const size_t lo0 = ((size_t)blockIdx.x % ntile_lo) << log2C, hi = (size_t)blockIdx.x / ntile_lo;
    const size_t base = (hi << (bl + B)) | lo0;
    const unsigned elems = (1u << B) << log2C;
    for (unsigned e = t; e < elems; e += T) {
        const unsigned mid = e >> log2C, c = e & (C - 1);
        const size_t g = base + ((size_t)mid << bl) + c;
        U x = Z::load(fft_load(a, g));
        if (pre != nullptr) x = Z::mul(x, fft_load(pre, pre_rev ? fft_bitrev(g, log2n) : g));
        tile[e] = x;
    }
    __syncthreads();
    const unsigned nbf = elems >> 1;
    const unsigned one_bit = TOPDOWN ? log2n - 1 : 0u;  // the stage whose twiddles are all one
    // one butterfly of the stage of tile bit bb: x = the element without the bit (global index i), y = the one with it
    auto butterfly = [&](U &x, U &y, unsigned bb, size_t i, bool carry) {
        const unsigned b = bl + bb;
        if (b == one_bit) {
            Z::dit_one(x, y);
        } else {
            const size_t tw = TOPDOWN ? (i >> (b + 1)) : ((i & (((size_t)1 << b) - 1)) << (log2n - 1 - b));
            if (carry) Z::template dit_free<true>(x, y, fft_load(twz, tw));
            else Z::template dit_free<false>(x, y, fft_load(twz, tw));
        }
    };
    for (unsigned st = 0; st < B; ++st) {
        const unsigned bb = TOPDOWN ? B - 1 - st : st;
        const bool carry = ((B - 1 - st) & 1u) != 0;  // every second stage, never the last of the pass (FftLz: LIMBS)
        for (unsigned q = t; q < nbf; q += T) {
            const unsigned c = q & (C - 1), p = q >> log2C;
            const unsigned mid0 = ((p >> bb) << (bb + 1)) | (p & ((1u << bb) - 1)), mid1 = mid0 | (1u << bb);
            U x = tile[(mid0 << log2C) + c], y = tile[(mid1 << log2C) + c];
            butterfly(x, y, bb, base + ((size_t)mid0 << bl) + c, carry);
            tile[(mid0 << log2C) + c] = x;
            tile[(mid1 << log2C) + c] = y;
        }
        __syncthreads();
    }
    for (unsigned e = t; e < elems; e += T) {
        const unsigned mid = e >> log2C, c = e & (C - 1);
        const size_t g = base + ((size_t)mid << bl) + c;
        U x = tile[e];
        if (post_mode != 0) {  // the transform's last pass: a product takes the tile's class (< 40q) to below 1.3q
            x = post_mode == 3 ? Z::mul(x, post_c) : Z::mul(x, fft_load(post, post_mode == 2 ? fft_bitrev(g, log2n) : g));
            fft_store(a, g, Z::store(x));
        } else {
            fft_store(a, g, final_pass ? Z::store_big(x) : Z::store_lazy_big(x));  // the next pass re-cuts whatever fits the element
        }
    }
}

// (Round 5 retried the same on the reduction-free lazy butterflies - the verdict's "radix 4"; the product count cannot drop, the fourth root
// of unity being a full product in a prime field -: 8 % slower at 2^24, profiles/r05_fft_fuse2.log; the code is in the history, commit "fr/fft: two
// stages per LDS round trip".)
// (Round 3 tried two radix-2 stages per trip through LDS - "radix 2^2": a thread owns the four elements that differ in
// two adjacent bits and runs both stages in registers, half the LDS round trips and barriers per butterfly. Slower at every
// size - 2^24 2.87 against 2.72 ms, 2^20 0.245 against 0.197 with 256 threads; 128 and 512 threads per tile are worse still
// (profiles/r03_fft_radix4.log): twice the registers (121 against 61) and a quarter of a tile's elements per work item
// leave fewer waves to cover the twiddle loads. Removed.)

// BitReverse (bitreverse.go:33-45): swap a[i] and a[rev(i)] once per pair
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_bit_reverse(Fp<FrP> *__restrict__ a, size_t n, unsigned log2n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t r = fft_bitrev(i, log2n);
    if (r > i) {
        const Fp<FrP> x = fft_load(a, i), y = fft_load(a, r);
        fft_store(a, i, y);
        fft_store(a, r, x);
    }
}

// ------------------------------------------------------------------ host side of one scalar field
// dynamic LDS above 64 KiB needs the attribute once per kernel (per device; set on whichever device is current)
static inline int ctx_allow_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : done)
        if (e.first == kernel && e.second == dev) return GMSM_OK;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.emplace_back(kernel, dev);
    return GMSM_OK;
}

template <class FrP>
struct FftField {
    using Fr = Fp<FrP>;
    static Fr from_words(const uint32_t *w) {
        Fr r;
        for (int i = 0; i < FrP::N; ++i) r.l[i] = w[i];
        return r;
    }
    static Fr pow2k(Fr x, unsigned k) {  // x^(2^k)
        for (unsigned i = 0; i < k; ++i) x = fp_sqr(x);
        return x;
    }
    static FftPowers<FrP> powers_of(Fr base) {
        FftPowers<FrP> pw;
        for (int b = 0; b < 40; ++b) {
            pw.p[b] = base;
            base = fp_sqr(base);
        }
        return pw;
    }

    // NewDomain (domain.go:66-99): cardinality 2^log2n, Generator = rootOfUnity^(2^(maxOrder - log2n)) (generator.go:32-34)
    static int domain_new(Context &ctx, hipStream_t stream, unsigned log2n, FftDomain *d) {
        (void)ctx;
        if (log2n > FrP::MAX_ORDER) return fail(GMSM_ERR_ARG, "m is too big: the required root of unity does not exist");
        const size_t n = (size_t)1 << log2n;
        d->log2n = log2n;
        const Fr gen = pow2k(from_words(FrP::ROOT_OF_UNITY), FrP::MAX_ORDER - log2n);
        const Fr gen_inv = fp_inv(gen);
        const Fr shift = from_words(FrP::MULT_GEN), shift_inv = fp_inv(shift);
        Fr card = Fr::one();  // n as a field element: Montgomery form of 2^log2n = ONE doubled log2n times
        for (unsigned i = 0; i < log2n; ++i) card = fp_dbl(card);
        const Fr card_inv = fp_inv(card);
        auto put = [&](std::vector<uint64_t> &dst, const Fr &v) {
            dst.resize(sizeof(Fr) / 8);
            memcpy(dst.data(), &v, sizeof(Fr));
        };
        put(d->generator, gen);
        put(d->generator_inv, gen_inv);
        put(d->cardinality_inv, card_inv);
        put(d->cardinality_inv_lz, fp_mul(card_inv, lazy_shift()));
        put(d->shift, shift);
        put(d->shift_inv, shift_inv);
        // twiddles w^t and w^-t, t < n/2 (preComputeTwiddles, domain.go:128-160, flattened to one table per direction),
        // in the lazy domain
        const size_t half = n / 2;
        int rc;
        if (half) {
            if ((rc = d->twiddles_lz.ensure(half * sizeof(Fr)))) return rc;
            if ((rc = d->twiddles_inv_lz.ensure(half * sizeof(Fr)))) return rc;
            const unsigned blocks = (unsigned)((half + 255) / 256);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen), lazy_shift(), half,
                               (Fr *)d->twiddles_lz.ptr, 0u);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen_inv), lazy_shift(), half,
                               (Fr *)d->twiddles_inv_lz.ptr, 0u);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        return GMSM_OK;
    }

    // 2^(L*W - 32N) as a field element: the factor every lazy-domain table entry carries (gmsm_fft_lazy.h)
    static Fr lazy_shift() {
        Fr s = Fr::one();
        for (unsigned i = 0; i < FftLz<FrP>::DOMAIN_SHIFT; ++i) s = fp_dbl(s);
        return s;
    }

    // The twiddle tables in bit-reversed order (entry k = w^(+-bitrev_(log n - 1)(k))): what the top-down passes (decimation
    // DIF) index with the block number. Built by the first such transform, like the coset tables; the caller holds d->mu.
    static int ensure_rev_tables(hipStream_t stream, FftDomain *d) {
        if (d->rev_ready) return GMSM_OK;
        const size_t half = ((size_t)1 << d->log2n) / 2;
        if (half) {
            int rc;
            if ((rc = d->twiddles_rev_lz.ensure(half * sizeof(Fr)))) return rc;
            if ((rc = d->twiddles_inv_rev_lz.ensure(half * sizeof(Fr)))) return rc;
            Fr gen, gen_inv;
            memcpy(&gen, d->generator.data(), sizeof(Fr));
            memcpy(&gen_inv, d->generator_inv.data(), sizeof(Fr));
            const unsigned blocks = (unsigned)((half + 255) / 256);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen), lazy_shift(), half,
                               (Fr *)d->twiddles_rev_lz.ptr, d->log2n - 1);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen_inv), lazy_shift(), half,
                               (Fr *)d->twiddles_inv_rev_lz.ptr, d->log2n - 1);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(stream));  // complete before the flag says so (any stream may read them next)
        }
        d->rev_ready = true;
        return GMSM_OK;
    }

    // cosetTable = u^i and cosetTableInv (with 1/n folded in) = u^-i / n, built on first use (domain.go:150-160); both in
    // the lazy domain
    static int ensure_coset_tables(hipStream_t stream, FftDomain *d) {
        if (d->coset_ready) return GMSM_OK;
        const size_t n = (size_t)1 << d->log2n;
        int rc;
        if ((rc = d->coset.ensure(n * sizeof(Fr)))) return rc;
        if ((rc = d->coset_inv_scaled.ensure(n * sizeof(Fr)))) return rc;
        Fr shift, shift_inv, card_inv_lz;
        memcpy(&shift, d->shift.data(), sizeof(Fr));
        memcpy(&shift_inv, d->shift_inv.data(), sizeof(Fr));
        memcpy(&card_inv_lz, d->cardinality_inv_lz.data(), sizeof(Fr));
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(shift), lazy_shift(), n, (Fr *)d->coset.ptr);
        hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(shift_inv), card_inv_lz, n,
                           (Fr *)d->coset_inv_scaled.ptr);
        HIP_TRY(hipGetLastError());
        // The tables are shared by every later transform on ANY stream (the caller holds d->mu only while it enqueues):
        // they must be complete before the flag says so, not merely queued on the builder's stream.
        HIP_TRY(hipStreamSynchronize(stream));
        d->coset_ready = true;
        return GMSM_OK;
    }

    // (*Domain).FFT / FFTInverse on a device vector of n = cardinality elements
    static int run(hipStream_t stream, FftDomain *d, void *d_a, bool inverse, bool dif, bool coset) {
        const unsigned log2n = d->log2n;
        const size_t n = (size_t)1 << log2n;
        Fr *a = (Fr *)d_a;
        const unsigned blocks_n = (unsigned)((n + 255) / 256);
        int rc;
        if (coset && (rc = ensure_coset_tables(stream, d))) return rc;
        if (dif && n > 1 && (rc = ensure_rev_tables(stream, d))) return rc;
        const bool fused = n > 1;  // the scalings ride on the first / last pass
        Fr card_inv_lz;
        memcpy(&card_inv_lz, d->cardinality_inv_lz.data(), sizeof(Fr));
        // the two scalings of the transform (fft.go:43-82, :144-195): DIT input and DIF output are bit-reversed, so the
        // tables are read in bit-reversed order there
        const Fr *pre = (coset && !inverse) ? (const Fr *)d->coset.ptr : nullptr;
        const int pre_rev = dif ? 0 : 1;
        const Fr *post = (inverse && coset) ? (const Fr *)d->coset_inv_scaled.ptr : nullptr;
        const int post_mode = !inverse ? 0 : coset ? (dif ? 2 : 1) : 3;
        if (pre && !fused)
            hipLaunchKernelGGL((k_fft_scale_table_lz<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, log2n, pre, pre_rev);
        if (n > 1) {
            // passes over the bit positions: the lowest LOWB bits as one contiguous pass (tiles of 2^LOWB elements,
            // C = 1), the rest in passes of <= 8 bits with C = 8 consecutive elements per row (2^24: 10 + 7 + 7). DIF runs the passes
            // from the top bits down, DIT from the bottom up.
            constexpr unsigned LOWB = sizeof(Fr) <= 32 ? GMSM_FFT_LOWB : GMSM_FFT_LOWB - 1;  // 2^10 x 36 B = 36 KiB of LDS
            const unsigned tpb = log2n >= 22 ? 512u : 256u;
            static_assert(LOWB <= FftLz<FrP>::DIT_FREE_STAGES - 1 && 8 <= FftLz<FrP>::DIT_FREE_STAGES - 1, "stages of a reduction-free pass");
            struct Pass { unsigned bl, B, log2C; } passes[16];
            int np = 0;
            const unsigned low = std::min(log2n, LOWB);
            passes[np++] = Pass{0, low, 0};
            unsigned rest = log2n - low, bl = low;
            const unsigned nhi = (rest + 7) / 8;
            for (unsigned k = 0; k < nhi; ++k) {
                const unsigned Bk = rest / (nhi - k) + ((rest % (nhi - k)) ? 1 : 0);
                passes[np++] = Pass{bl, Bk, 3};
                bl += Bk;
                rest -= Bk;
            }
            {
                const Fr *twz = dif ? (const Fr *)(inverse ? d->twiddles_inv_rev_lz.ptr : d->twiddles_rev_lz.ptr)
                                    : (const Fr *)(inverse ? d->twiddles_inv_lz.ptr : d->twiddles_lz.ptr);
                for (int k = 0; k < np; ++k) {
                    const Pass &ps = passes[dif ? np - 1 - k : k];
                    const size_t lds = ((size_t)sizeof(FpU<FrP>) << ps.B) << ps.log2C;
                    const size_t tiles = n >> (ps.B + ps.log2C);
                    const Fr *pre_k = k == 0 ? pre : nullptr;
                    const int post_k = k == np - 1 ? post_mode : 0;
                    if (dif) {
                        if ((rc = ctx_allow_lds((const void *)k_fft_pass_lz<FrP, true>, 128 * 1024))) return rc;
                        hipLaunchKernelGGL((k_fft_pass_lz<FrP, true>), dim3((unsigned)tiles), dim3(tpb), lds, stream, a, log2n, ps.bl,
                                           ps.B, ps.log2C, twz, pre_k, pre_rev, post, post_k, card_inv_lz, k == np - 1 ? 1 : 0);
                    } else {
                        if ((rc = ctx_allow_lds((const void *)k_fft_pass_lz<FrP, false>, 128 * 1024))) return rc;
                        hipLaunchKernelGGL((k_fft_pass_lz<FrP, false>), dim3((unsigned)tiles), dim3(tpb), lds, stream, a, log2n, ps.bl,
                                           ps.B, ps.log2C, twz, pre_k, pre_rev, post, post_k, card_inv_lz, k == np - 1 ? 1 : 0);
                    }
                }
            }
        }
        if (inverse && !fused) {
            if (!coset) hipLaunchKernelGGL((k_fft_scale_const_lz<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, card_inv_lz);
            else hipLaunchKernelGGL((k_fft_scale_table_lz<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, log2n, post, post_mode == 2 ? 1 : 0);
        }
        HIP_TRY(hipGetLastError());
        return GMSM_OK;
    }

    static int bit_reverse(hipStream_t stream, void *d_a, size_t n) {
        unsigned log2n = 0;
        while (((size_t)1 << log2n) < n) ++log2n;
        if (((size_t)1 << log2n) != n) return fail(GMSM_ERR_ARG, "len(a) must be a power of 2");
        if (n > 1)
            hipLaunchKernelGGL((k_fft_bit_reverse<FrP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (Fr *)d_a, n, log2n);
        HIP_TRY(hipGetLastError());
        return GMSM_OK;
    }
};

}  // namespace gmsm
