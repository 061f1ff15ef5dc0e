// The function table of one (curve, group): what the C ABI (gmsm_engine.hip) dispatches through - argument marshalling
// between the ABI's uint64_t limbs / void pointers and the typed entry points of struct Group. Included by gmsm_group.h.
#pragma once
#include "gmsm_group.h"
#include "gmsm_group_debug.h"

namespace gmsm {

// ------------------------------------------------------------------ function table

template <class G>
struct VTableOf {
    static int multiexp_host(const uint64_t *points, size_t n_points, const uint64_t *scalars, size_t n_scalars,
                             int nb_tasks, uint64_t *out_jac) {
        typename G::J j;
        int rc = G::multiexp_host(points, n_points, scalars, n_scalars, nb_tasks, &j);
        if (rc) return rc;
        memcpy(out_jac, &j, sizeof j);
        return GMSM_OK;
    }
    static int multiexp_device(Context &ctx, const void *d_points, const void *d_scalars, size_t n, hipStream_t stream,
                               uint64_t *out_jac, const ResidentBases *resident) {
        typename G::J j;
        if (n == 0) j = typename G::J{G::F::one(), G::F::one(), G::F::zero()};
        else {
            GMSM_LEASE_OR_FAIL(lease, ctx);
            int rc = G::multiexp_device(ctx, *lease.w, d_points, d_scalars, n, stream, &j, resident);
            if (rc) return rc;
        }
        memcpy(out_jac, &j, sizeof j);
        return GMSM_OK;
    }
    static int multiexp_bases_host(Context &ctx, const uint64_t *scalars, size_t n, uint64_t *out_jac,
                                   const ResidentBases *resident) {
        GMSM_LEASE_OR_FAIL(lease, ctx);
        typename G::J j;
        int rc = G::multiexp_from_host(ctx, *lease.w, nullptr, resident, scalars, n, &j);
        if (rc) return rc;
        memcpy(out_jac, &j, sizeof j);
        return GMSM_OK;
    }
    static int shard_piece(Context &ctx, const uint64_t *points, const ResidentBases *resident, size_t resident_base,
                           const uint64_t *scalars, size_t n, unsigned c, unsigned win_first, unsigned win_stride,
                           uint64_t *out_xyzz) {
        return G::shard_piece(ctx, points, resident, resident_base, scalars, n, c, win_first, win_stride,
                              reinterpret_cast<typename G::Ext *>(out_xyzz));
    }
    static unsigned host_piece_ranges(size_t n, bool with_points) { return G::host_range_count(n, with_points, G::plan_for(nullptr, n)); }
    static int precompute_tables(Context &ctx, Workspace &ws, ResidentBases *rb, unsigned c) {
        return G::precompute_tables(ctx, ws, rb, c);
    }
    static bool tables_serve(size_t n_registered, size_t n_call) { return G::tables_serve(n_registered, n_call); }
    static int window_sums(Context &ctx, const void *d_points, const void *d_scalars, size_t n, unsigned c,
                           unsigned win_first, unsigned win_stride, hipStream_t stream, uint64_t *out_xyzz,
                           const ResidentBases *resident) {
        WindowPlan plan = G::make_plan(c, win_first, win_stride);
        GMSM_LEASE_OR_FAIL(lease, ctx);
        return G::window_sums(ctx, *lease.w, d_points, d_scalars, n, plan, stream,
                              reinterpret_cast<typename G::Ext *>(out_xyzz), resident);
    }
    static int register_bases(Context &ctx, const void *d_points, size_t n, hipStream_t stream, ResidentBases *out) {
        return G::register_bases(ctx, d_points, n, stream, out);
    }
    static int window_sums_enqueue(Context &ctx, const void *d_points, const void *d_scalars, size_t n, unsigned c,
                                   unsigned win_first, unsigned win_stride, hipStream_t stream, void *d_out_xyzz,
                                   const std::shared_ptr<ResidentBases> &resident) {
        WindowPlan plan = G::make_plan(c, win_first, win_stride);
        // leased for the duration of this call only: the work it leaves in flight is protected by stream order
        // (Workspace::last_use, honoured by whoever leases the workspace next), not by the lease
        GMSM_LEASE_OR_FAIL(lease, ctx);
        Workspace *ws = lease.w;
        int rc = G::enqueue_window_sums(ctx, *ws, d_points, d_scalars, n, plan, stream, resident.get(), d_out_xyzz);
        ws->bases_ref = resident;             // alive until the next call on this workspace replaces it
        ws->uncollected = ws->pending_timed;  // nobody waits for this call: its stage events are read by the next one
        ws->pending_timed = false;
        return rc;
    }
    static void fold_sets(const uint64_t *xyzz_sets, unsigned nsets, unsigned c, uint64_t *out_jac) {
        typename G::J j = G::fold_sets(reinterpret_cast<const typename G::Ext *>(xyzz_sets), nsets, c);
        memcpy(out_jac, &j, sizeof j);
    }
    static void fold_powers(const uint64_t *coeff, size_t n, uint64_t *out_scalars) { G::fold_powers(coeff, n, out_scalars); }
    // host or device scalars / results; exactly one of each pair is given
    static int batch_scalar_mul(Context &ctx, const uint64_t *base, const uint64_t *scalars, const void *d_scalars, size_t n,
                                hipStream_t caller_stream, uint64_t *out, void *d_out) {
        GMSM_LEASE_OR_FAIL(lease, ctx);
        Workspace &ws = *lease.w;
        int rc = order_after(ws, caller_stream);
        if (rc) return rc;
        const void *dsc = d_scalars;
        if (scalars && n) {
            if ((rc = ws.h2d_scalars.ensure(n * G::SCALAR_BYTES))) return rc;
            HIP_TRY(hipMemcpyAsync(ws.h2d_scalars.ptr, scalars, n * G::SCALAR_BYTES, hipMemcpyHostToDevice, ws.stream));
            dsc = ws.h2d_scalars.ptr;
        }
        void *dres = d_out;
        if (out && n) {
            if ((rc = ws.parted.ensure(n * G::AFF_BYTES))) return rc;
            dres = ws.parted.ptr;
        }
        if ((rc = G::batch_scalar_mul(ctx, ws, base, dsc, n, dres))) return rc;
        if (out && n) HIP_TRY(hipMemcpyAsync(out, dres, n * G::AFF_BYTES, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
        return GMSM_OK;
    }
    static int batch_jac_to_affine(Context &ctx, const uint64_t *jac, size_t n, uint64_t *out) {
        if (n == 0) return GMSM_OK;
        GMSM_LEASE_OR_FAIL(lease, ctx);
        Workspace &ws = *lease.w;
        int rc;
        if ((rc = ws.h2d_points.ensure(n * sizeof(typename G::J)))) return rc;
        if ((rc = ws.parted.ensure(n * G::AFF_BYTES))) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h2d_points.ptr, jac, n * sizeof(typename G::J), hipMemcpyHostToDevice, ws.stream));
        if ((rc = G::batch_jac_to_affine(ws, ws.h2d_points.ptr, n, ws.parted.ptr))) return rc;
        HIP_TRY(hipMemcpyAsync(out, ws.parted.ptr, n * G::AFF_BYTES, hipMemcpyDeviceToHost, ws.stream));
        HIP_TRY(hipStreamSynchronize(ws.stream));
        return GMSM_OK;
    }
    static int submit(Context &ctx, Workspace &ws, const void *d_scalars, size_t n, const ResidentBases *resident) {
        return G::multiexp_submit(ctx, ws, d_scalars, n, resident);
    }
    static int decode_raw(Workspace &ws, const void *d_raw, size_t n, int level, void *d_out, long long *bad_index,
                          uint32_t *status) {
        return G::decode_raw(ws, d_raw, n, level, d_out, bad_index, status);
    }
    static int decode_compressed(Workspace &ws, const void *d_comp, size_t n, int level, void *d_out, long long *bad_index,
                                 uint32_t *status) {
        return G::decode_compressed(ws, d_comp, n, level, d_out, bad_index, status);
    }
    static int encode_compressed(Workspace &ws, const void *d_points, size_t n, void *d_comp) {
        return G::encode_compressed(ws, d_points, n, d_comp);
    }
    static int fft_domain_new(Context &ctx, hipStream_t stream, unsigned log2n, FftDomain *out) {
        return FftField<typename G::FrP>::domain_new(ctx, stream, log2n, out);
    }
    static int fft_run(hipStream_t stream, FftDomain *d, void *d_a, bool inverse, bool dif, bool coset) {
        return FftField<typename G::FrP>::run(stream, d, d_a, inverse, dif, coset);
    }
    static int fft_bit_reverse(hipStream_t stream, void *d_a, size_t n) {
        return FftField<typename G::FrP>::bit_reverse(stream, d_a, n);
    }
    static int validate_points(Workspace &ws, const void *d_points, size_t n, int level, long long *bad_index,
                               uint32_t *status) {
        return G::validate_points(ws, d_points, n, level, bad_index, status);
    }
    static int collect(Workspace &ws, uint64_t *out_jac) {
        typename G::J j;
        int rc = G::multiexp_collect(ws, &j);
        if (rc) return rc;
        memcpy(out_jac, &j, sizeof j);
        return GMSM_OK;
    }
    static void fold(const uint64_t *xyzz_windows, unsigned c, uint64_t *out_jac) {
        typename G::J j = G::fold(reinterpret_cast<const typename G::Ext *>(xyzz_windows), c);
        memcpy(out_jac, &j, sizeof j);
    }
    static void jac_to_affine(const uint64_t *jac, uint64_t *out_affine) {
        typename G::J j;
        memcpy(&j, jac, sizeof j);
        typename G::Aff a = affine_from_jac(j);
        memcpy(out_affine, &a, sizeof a);
    }
    static int debug_decompose(const uint64_t *scalars, size_t n, unsigned c, uint32_t *out_digits) {
        return debug_decompose_impl<G>(scalars, n, c, out_digits);
    }
    static int debug_glv_split(const uint64_t *scalars, size_t n, uint32_t *out) { return debug_glv_split_impl<G>(scalars, n, out); }
    // what a MultiExp over n bases taken anew runs as (gmsm_default_plan)
    static void plan_info(size_t n, unsigned *c, unsigned *nwin, unsigned *entries_per_point, unsigned *fused) {
        if (G::small_serves(n, nullptr)) {
            const typename G::SmallPlan sp = G::small_plan(n, nullptr);
            *c = sp.plan.c, *nwin = sp.plan.nwin_total, *entries_per_point = sp.glv ? 2u : 1u, *fused = 1u;
            return;
        }
        const WindowPlan p = G::plan_for(nullptr, n);
        *c = p.c, *nwin = p.nwin_total, *entries_per_point = p.glv ? 2u : 1u, *fused = 0u;
    }
    static int debug_field_op(int field, int op, const uint64_t *a, const uint64_t *b, size_t count, uint64_t *out) {
        using BaseP = typename G::F::Params;
        if (field == 0) {
            if (op == 6) return debug_from_mont<BaseP>(a, count, out);
            return debug_field<Fp<BaseP>>(op, a, b, count, out);
        } else if (field == 1) {
            if (op == 6) return debug_from_mont<typename G::FrP>(a, count, out);
            return debug_field<Fp<typename G::FrP>>(op, a, b, count, out);
        }
        if (op == 6) return fail(GMSM_ERR_ARG, "from_mont is defined on prime fields only");
        if (field == 3) {  // coordinate field through the lazy-limb code
            using U = typename G::U;
            using S = typename LzTraits<U>::Sat;
            return run_elementwise<S>(count * sizeof(S), a, b ? count * sizeof(S) : 0, b, count * sizeof(S), out,
                                      [&](void *da, void *db, void *dout, hipStream_t s) {
                                          hipLaunchKernelGGL((k_lazy_field_op<U>), dim3((unsigned)((count + 63) / 64)), dim3(64),
                                                             0, s, op, (const S *)da, (const S *)db, count, (S *)dout);
                                      });
        }
        return debug_field<typename G::F>(op, a, b, count, out);
    }
    static int debug_group_op(int op, const uint64_t *acc, const uint64_t *other, size_t count, uint64_t *out) {
        if (op >= 4) {  // 4..7 = ops 0..3 through the lazy-limb group law
            using U = typename G::U;
            using F = typename G::F;
            const int lop = op - 4;
            const size_t ob = (lop == 0 || lop == 1) ? sizeof(Affine<F>) : sizeof(XYZZ<F>);
            return run_elementwise<F>(count * sizeof(XYZZ<F>), acc, other ? count * ob : 0, other, count * sizeof(XYZZ<F>), out,
                                      [&](void *da, void *db, void *dout, hipStream_t s) {
                                          hipLaunchKernelGGL((k_lazy_group_op<U>), dim3((unsigned)((count + 63) / 64)), dim3(64),
                                                             0, s, lop, (const XYZZ<F> *)da, (const void *)db, count,
                                                             (XYZZ<F> *)dout);
                                      });
        }
        return debug_group<G>(op, acc, other, count, out);
    }
    static void generate_points(const uint64_t *base, const uint64_t *k0, const uint64_t *k1, int klimbs, size_t n,
                                int nthreads, uint64_t *out) {
        G::generate_points(base, k0, k1, klimbs, n, nthreads, out);
    }
    static const GroupVTable *get() {
        static const GroupVTable vt = {G::FR_BITS,      G::AFF_BYTES,   G::SCALAR_BYTES, sizeof(typename G::J),
                                       sizeof(typename G::Ext), &multiexp_host, &multiexp_device, &window_sums,
                                       &fold,           &jac_to_affine, &debug_decompose, &debug_field_op,
                                       &debug_group_op, &generate_points, &register_bases, &submit, &collect, &window_sums_enqueue, &fold_sets, &fold_powers, &multiexp_bases_host, &batch_scalar_mul, &batch_jac_to_affine, &decode_raw, &validate_points, &decode_compressed, &encode_compressed, &fft_domain_new, &fft_run, &fft_bit_reverse, &precompute_tables, &tables_serve, &shard_piece, &host_piece_ranges, &debug_glv_split, &plan_info};
        return &vt;
    }
};

}  // namespace gmsm
