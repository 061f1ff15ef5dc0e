// Test hooks of one (curve, group): field / group operations and the decomposition run on the device for the parity tests
// (gmsm_debug_* in include/gmsm.h; tests/test_gpu_parity.py). Included by gmsm_group.h, after struct Group.
#pragma once
#include "gmsm_group.h"

namespace gmsm {

// ------------------------------------------------------------------ debug / test kernels
template <class FT>
__global__ void k_field_op(int op, const FT *a, const FT *b, size_t count, FT *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    FT x = a[i], y = b ? b[i] : a[i], z;
    switch (op) {
        case 0: z = fp_mul(x, y); break;
        case 1: z = fp_add(x, y); break;
        case 2: z = fp_sub(x, y); break;
        case 3: z = fp_neg(x); break;
        case 4: z = fp_dbl(x); break;
        default: z = fp_sqr(x); break;
    }
    out[i] = z;
}

template <class P>
__global__ void k_from_mont(const Fp<P> *a, size_t count, Fp<P> *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = fp_from_mont(a[i]);
}

template <class F>
__global__ void k_group_op(int op, const XYZZ<F> *acc, const void *other, size_t count, XYZZ<F> *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    XYZZ<F> p = acc[i];
    if (op == 0 || op == 1) {
        Affine<F> a = reinterpret_cast<const Affine<F> *>(other)[i];
        xyzz_add_mixed(p, a, op == 1);
    } else if (op == 2) {
        XYZZ<F> q = reinterpret_cast<const XYZZ<F> *>(other)[i];
        xyzz_add(p, q);
    } else {
        p = xyzz_double(p);
    }
    out[i] = p;
}

// ---- the same hooks through the lazy-limb code paths (gmsm_fieldu.h / gmsm_field2u.h / gmsm_curveu.h): operands are
// converted from canonical saturated limbs, the lazy operation runs, the result is converted back. This exercises
// pack/unpack, the carry passes, the redundant-constant subtractions, the conditional reductions and the flag-based
// infinity handling on edge values, independently of the MSM pipeline.
template <class P> __device__ __forceinline__ FpU<P> dbg_add(const FpU<P> &a, const FpU<P> &b) { return fpu_add(a, b); }
template <class P> __device__ __forceinline__ FpU<P> dbg_sub(const FpU<P> &a, const FpU<P> &b) { return fpu_sub<P, 4>(a, b); }
template <class P> __device__ __forceinline__ FpU<P> dbg_neg(const FpU<P> &a) { return fpu_neg4<P>(a); }
template <class P> __device__ __forceinline__ FpU<P> dbg_dbl(const FpU<P> &a) { return fpu_dbl(a); }
template <class P> __device__ __forceinline__ Fp2U<P> dbg_add(const Fp2U<P> &a, const Fp2U<P> &b) { return lz_add(a, b); }
template <class P> __device__ __forceinline__ Fp2U<P> dbg_sub(const Fp2U<P> &a, const Fp2U<P> &b) { return lz_sub(a, b); }
template <class P> __device__ __forceinline__ Fp2U<P> dbg_neg(const Fp2U<P> &a) { return lz_sub(lz_zero((const Fp2U<P> *)nullptr), a); }
template <class P> __device__ __forceinline__ Fp2U<P> dbg_dbl(const Fp2U<P> &a) { return lz_dbl(a); }

template <class U>
__global__ void k_lazy_field_op(int op, const typename LzTraits<U>::Sat *a, const typename LzTraits<U>::Sat *b, size_t count,
                                typename LzTraits<U>::Sat *out) {
    using T = LzTraits<U>;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const U x = T::template from_sat<true>(a[i]);
    const U y = T::template from_sat<true>(b ? b[i] : a[i]);
    U z;
    switch (op) {
        case 0: z = lz_mul<true>(x, y); break;
        case 1: z = dbg_add(x, y); break;
        case 2: z = dbg_sub(x, y); break;
        case 3: z = dbg_neg(x); break;
        case 4: z = dbg_dbl(x); break;
        case 7:  // a square root (gmsm_decompress.h), zero when there is none (the tests give non-zero operands)
            if (!lz_sqrt(x, z)) z = lz_zero((const U *)nullptr);
            break;
        default: z = lz_sqr<true>(x); break;
    }
    out[i] = T::template to_sat<true>(z);
}

template <class U>
__global__ void k_lazy_group_op(int op, const XYZZ<typename LzTraits<U>::Sat> *acc, const void *other, size_t count,
                                XYZZ<typename LzTraits<U>::Sat> *out) {
    using T = LzTraits<U>;
    using S = typename T::Sat;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    auto to_lazy = [](const XYZZ<S> &m) {
        UnsatElem<U> e;
        e.inf = m.zz.is_zero();
        e.v.x = T::template from_sat<true>(m.x);
        e.v.y = T::template from_sat<true>(m.y);
        e.v.zz = T::template from_sat<true>(m.zz);
        e.v.zzz = T::template from_sat<true>(m.zzz);
        return e;
    };
    UnsatElem<U> p = to_lazy(acc[i]);
    if (op == 0 || op == 1) {
        const Affine<S> a = reinterpret_cast<const Affine<S> *>(other)[i];
        if (!a.is_infinity()) {  // the pipeline drops points at infinity before the accumulation (k_decompose skip flags)
            lz_madd_acc<true>(p.v, p.inf, T::template from_sat<true>(a.x), T::template from_sat<true>(a.y), op == 1);
            lz_acc_finish(p.v, p.inf);
        }
    } else if (op == 2) {
        const UnsatElem<U> q = to_lazy(reinterpret_cast<const XYZZ<S> *>(other)[i]);
        lz_padd<true>(p.v, p.inf, q.v, q.inf);
    } else if (!p.inf) {
        p.v = lz_pdbl<true>(p.v);
    }
    unsat_store_final<U, true>(out, i, p);
}

template <class FT, class Launch>
static int run_elementwise(size_t in_bytes_a, const void *a, size_t in_bytes_b, const void *b, size_t out_bytes, void *out,
                           Launch launch) {
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    hipStream_t stream = lease.w->stream;
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&da, in_bytes_a));
    HIP_TRY(hipMemcpy(da, a, in_bytes_a, hipMemcpyHostToDevice));
    if (b) {
        HIP_TRY(hipMalloc(&db, in_bytes_b));
        HIP_TRY(hipMemcpy(db, b, in_bytes_b, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&dout, out_bytes));
    launch(da, db, dout, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    if (db) (void)hipFree(db);
    (void)hipFree(dout);
    return GMSM_OK;
}

template <class FT>
static int debug_field(int op, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out) {
    return run_elementwise<FT>(count * sizeof(FT), a, b ? count * sizeof(FT) : 0, b, count * sizeof(FT), out,
                               [&](void *da, void *db, void *dout, hipStream_t s) {
                                   hipLaunchKernelGGL((k_field_op<FT>), dim3((unsigned)((count + 127) / 128)), dim3(128), 0,
                                                      s, op, (const FT *)da, (const FT *)db, count, (FT *)dout);
                               });
}

template <class P>
static int debug_from_mont(const uint64_t *a, size_t count, uint64_t *out) {
    return run_elementwise<Fp<P>>(count * sizeof(Fp<P>), a, 0, nullptr, count * sizeof(Fp<P>), out,
                                  [&](void *da, void *, void *dout, hipStream_t s) {
                                      hipLaunchKernelGGL((k_from_mont<P>), dim3((unsigned)((count + 127) / 128)),
                                                         dim3(128), 0, s, (const Fp<P> *)da, count, (Fp<P> *)dout);
                                  });
}

template <class G>
static int debug_group(int op, const uint64_t *acc, const uint64_t *other, size_t count, uint64_t *out) {
    using F = typename G::F;
    const size_t ob = (op == 0 || op == 1) ? sizeof(Affine<F>) : sizeof(XYZZ<F>);
    return run_elementwise<F>(count * sizeof(XYZZ<F>), acc, other ? count * ob : 0, other, count * sizeof(XYZZ<F>), out,
                              [&](void *da, void *db, void *dout, hipStream_t s) {
                                  hipLaunchKernelGGL((k_group_op<F>), dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s,
                                                     op, (const XYZZ<F> *)da, (const void *)db, count, (XYZZ<F> *)dout);
                              });
}

template <class G>
static int debug_decompose_impl(const uint64_t *scalars, size_t n, unsigned c, uint32_t *out_digits) {
    if (c < 2 || c > 24) return fail(GMSM_ERR_ARG, "c out of range");
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    WindowPlan plan = G::make_plan(c, 0, 1);
    if (n == 0) return GMSM_OK;
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    if ((rc = ws.h2d_scalars.ensure(n * G::SCALAR_BYTES))) return rc;
    if ((rc = ws.digits.ensure((size_t)plan.nwin_total * n * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, scalars, n * G::SCALAR_BYTES, hipMemcpyHostToDevice, ws.stream));
    G::launch_decompose(ws.h2d_scalars.ptr, n, plan, /*d16=*/false, ws.digits.ptr, nullptr, ws.stream);  // the kernel the pipeline runs for this c
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_digits, ws.digits.ptr, (size_t)plan.nwin_total * n * 4, hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return GMSM_OK;
}


template <class G>
static int debug_glv_split_impl(const uint64_t *scalars, size_t n, uint32_t *out) {
    using FrP = typename G::FrP;
    Context *ctx;
    int rc = get_context(&ctx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return GMSM_OK;
    GMSM_LEASE_OR_FAIL(lease, *ctx);
    Workspace &ws = *lease.w;
    const size_t ob = n * 2 * (FrP::GLV_HL + 1) * 4;
    if ((rc = ws.h2d_scalars.ensure(n * G::SCALAR_BYTES))) return rc;
    if ((rc = ws.digits.ensure(ob))) return rc;
    HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, scalars, n * G::SCALAR_BYTES, hipMemcpyHostToDevice, ws.stream));
    hipLaunchKernelGGL((k_glv_split_debug<FrP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ws.stream,
                       (const uint32_t *)ws.h2d_scalars.ptr, n, (uint32_t *)ws.digits.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ws.digits.ptr, ob, hipMemcpyDeviceToHost, ws.stream));
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return GMSM_OK;
}

}  // namespace gmsm
