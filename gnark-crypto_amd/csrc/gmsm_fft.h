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
// Shape on the GPU: the reference recurses over halves with one goroutine per half; here every radix-2 stage is one
// launch over all n/2 butterflies (coalesced 32-byte elements, twiddles w^t for t < n/2 resident per domain like the
// reference's precomputed tables). The arithmetic is the canonical saturated Montgomery field (gmsm_field.h), so every
// intermediate equals the reference's and the result needs no normalisation.
#pragma once
#include <hip/hip_runtime.h>
#include "gmsm_context.h"
#include "gmsm_field.h"

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

// out[i] = scale * base^i, i < count (BuildExpTable, fft/domain.go; here one thread per entry)
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_pow_table(FftPowers<FrP> pw, Fp<FrP> scale, size_t count, Fp<FrP> *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fp<FrP> acc = scale;
#pragma nounroll
    for (int b = 0; b < 40; ++b)
        if ((i >> b) & 1) acc = fp_mul(acc, pw.p[b]);
    fft_store(out, i, acc);
}

__device__ __forceinline__ size_t fft_bitrev(size_t i, unsigned log2n) {
    return log2n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log2n)) : 0;
}

// a[i] *= table[rev ? bitrev(i) : i]   (coset scaling; table may carry 1/n folded in)
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_scale_table(Fp<FrP> *__restrict__ a, size_t n, unsigned log2n,
                                                         const Fp<FrP> *__restrict__ table, int rev) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t t = rev ? fft_bitrev(i, log2n) : i;
    fft_store(a, i, fp_mul(fft_load(a, i), fft_load(table, t)));
}

// a[i] *= c   (CardinalityInv, fft.go:144-150)
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_scale_const(Fp<FrP> *__restrict__ a, size_t n, Fp<FrP> c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fft_store(a, i, fp_mul(fft_load(a, i), c));
}

// One decimation-in-frequency stage (difFFT, fft.go:198-262): stage s works on blocks of 2*half, half = n >> (s+1):
//   (a[i], a[i+half]) <- (a[i] + a[i+half], (a[i] - a[i+half]) * w^(j << s)),  j = i mod half.
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_dif_stage(Fp<FrP> *__restrict__ a, size_t n, unsigned log2n, unsigned s,
                                                       const Fp<FrP> *__restrict__ tw) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n / 2) return;
    const unsigned lh = log2n - 1 - s;  // log2(half)
    const size_t half = (size_t)1 << lh;
    const size_t j = idx & (half - 1), i = ((idx >> lh) << (lh + 1)) + j;
    const Fp<FrP> x = fft_load(a, i), y = fft_load(a, i + half);
    fft_store(a, i, fp_add(x, y));
    Fp<FrP> d = fp_sub(x, y);
    if (j) d = fp_mul(d, fft_load(tw, j << s));  // w^0 = 1 (innerDIFWithTwiddles skips it the same way)
    fft_store(a, i + half, d);
}

// One decimation-in-time stage (ditFFT, fft.go:264-330): half = 1 << s,
//   t = a[i+half] * w^(j * n / (2 half));  (a[i], a[i+half]) <- (a[i] + t, a[i] - t).
template <class FrP>
__global__ void __launch_bounds__(256) k_fft_dit_stage(Fp<FrP> *__restrict__ a, size_t n, unsigned log2n, unsigned s,
                                                       const Fp<FrP> *__restrict__ tw) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n / 2) return;
    const size_t half = (size_t)1 << s;
    const size_t j = idx & (half - 1), i = ((idx >> s) << (s + 1)) + j;
    const Fp<FrP> x = fft_load(a, i);
    Fp<FrP> t = fft_load(a, i + half);
    if (j) t = fp_mul(t, fft_load(tw, j << (log2n - 1 - s)));
    fft_store(a, i, fp_add(x, t));
    fft_store(a, i + half, fp_sub(x, t));
}

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
        Fr card = Fr::zero();  // n as a field element: Montgomery form of 2^log2n = ONE doubled log2n times
        card = Fr::one();
        for (unsigned i = 0; i < log2n; ++i) card = fp_dbl(card);
        const Fr card_inv = fp_inv(card);
        auto put = [&](std::vector<uint64_t> &dst, const Fr &v) {
            dst.resize(sizeof(Fr) / 8);
            memcpy(dst.data(), &v, sizeof(Fr));
        };
        put(d->generator, gen);
        put(d->generator_inv, gen_inv);
        put(d->cardinality_inv, card_inv);
        put(d->shift, shift);
        put(d->shift_inv, shift_inv);
        // twiddles w^t and w^-t, t < n/2 (preComputeTwiddles, domain.go:128-160, flattened to one table per direction)
        const size_t half = n / 2;
        int rc;
        if (half) {
            if ((rc = d->twiddles.ensure(half * sizeof(Fr)))) return rc;
            if ((rc = d->twiddles_inv.ensure(half * sizeof(Fr)))) return rc;
            const unsigned blocks = (unsigned)((half + 255) / 256);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen), Fr::one(), half,
                               (Fr *)d->twiddles.ptr);
            hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(gen_inv), Fr::one(), half,
                               (Fr *)d->twiddles_inv.ptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        return GMSM_OK;
    }

    // cosetTable = u^i and cosetTableInv (with 1/n folded in) = u^-i / n, built on first use (domain.go:150-160)
    static int ensure_coset_tables(hipStream_t stream, FftDomain *d) {
        if (d->coset_ready) return GMSM_OK;
        const size_t n = (size_t)1 << d->log2n;
        int rc;
        if ((rc = d->coset.ensure(n * sizeof(Fr)))) return rc;
        if ((rc = d->coset_inv_scaled.ensure(n * sizeof(Fr)))) return rc;
        Fr shift, shift_inv, card_inv;
        memcpy(&shift, d->shift.data(), sizeof(Fr));
        memcpy(&shift_inv, d->shift_inv.data(), sizeof(Fr));
        memcpy(&card_inv, d->cardinality_inv.data(), sizeof(Fr));
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(shift), Fr::one(), n, (Fr *)d->coset.ptr);
        hipLaunchKernelGGL((k_fft_pow_table<FrP>), dim3(blocks), dim3(256), 0, stream, powers_of(shift_inv), card_inv, n,
                           (Fr *)d->coset_inv_scaled.ptr);
        HIP_TRY(hipGetLastError());
        d->coset_ready = true;
        return GMSM_OK;
    }

    // (*Domain).FFT / FFTInverse on a device vector of n = cardinality elements
    static int run(hipStream_t stream, FftDomain *d, void *d_a, bool inverse, bool dif, bool coset) {
        const unsigned log2n = d->log2n;
        const size_t n = (size_t)1 << log2n;
        Fr *a = (Fr *)d_a;
        const unsigned blocks_n = (unsigned)((n + 255) / 256), blocks_h = (unsigned)((n / 2 + 255) / 256);
        int rc;
        if (coset && (rc = ensure_coset_tables(stream, d))) return rc;
        if (coset && !inverse)  // fft.go:43-82: DIT input is bit-reversed, so the table is read in bit-reversed order
            hipLaunchKernelGGL((k_fft_scale_table<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, log2n, (const Fr *)d->coset.ptr,
                               dif ? 0 : 1);
        const Fr *tw = (const Fr *)(inverse ? d->twiddles_inv.ptr : d->twiddles.ptr);
        if (n > 1) {
            if (dif)
                for (unsigned s = 0; s < log2n; ++s)
                    hipLaunchKernelGGL((k_fft_dif_stage<FrP>), dim3(blocks_h), dim3(256), 0, stream, a, n, log2n, s, tw);
            else
                for (unsigned s = 0; s < log2n; ++s)
                    hipLaunchKernelGGL((k_fft_dit_stage<FrP>), dim3(blocks_h), dim3(256), 0, stream, a, n, log2n, s, tw);
        }
        if (inverse) {
            if (!coset) {
                Fr card_inv;
                memcpy(&card_inv, d->cardinality_inv.data(), sizeof(Fr));
                hipLaunchKernelGGL((k_fft_scale_const<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, card_inv);
            } else {  // fft.go:153-195: DIT output is natural, DIF output bit-reversed
                hipLaunchKernelGGL((k_fft_scale_table<FrP>), dim3(blocks_n), dim3(256), 0, stream, a, n, log2n,
                                   (const Fr *)d->coset_inv_scaled.ptr, dif ? 1 : 0);
            }
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
