// Lane-cooperative extended-Jacobian arithmetic with the operands in LDS ("quad" = 4 adjacent lanes of a wavefront).
//
// The tail of the bucket reduction (multiexp_jacobian.go:44-52 restricted to partial sums) is a short dependency chain
// of XYZZ additions on few elements: a lane that runs one addition alone issues its 14 field products back to back (8.6 us
// for BN254 G1, 43 us for the 28-limb field) while most lanes of the chip have nothing to do. An addition (add-2008-s,
// g1.go:736-788) has 14 products in 4 dependency levels of at most 4 independent products; four lanes that each take ONE
// product per level finish it in 4 product-times.
//
// Round 2's quad code kept full copies of both operands in every lane (8 field elements + the intermediates: over 500
// registers for the 28-limb field, 2.3 KB of scratch per lane, 66 us per step). Here the operands live in LDS records and
// every lane reads only the two field elements its own product needs; intermediates stay in the lane that produced them
// and travel by ds_bpermute only where another role needs them:
//
//   level   lane 0                 lane 1                 lane 2                   lane 3
//   1       U2 = Y.x  X.zz         U1 = X.x  Y.zz         S2 = Y.y  X.zzz          S1 = X.y  Y.zzz
//           A = U2 - U1(<-1)                              B = S2 - S1(<-3)
//   2       PP = A A               T2 = X.zzz Y.zzz       BB = B B                 T1 = X.zz Y.zz
//   3       PPP = A PP             Q = U1 PP(<-0)         -                        ZZ3 = T1 PP(<-0)
//                                                         X3 = BB - PPP(<-0) - 2Q(<-1)
//   4       -                      ZZZ3 = T2 PPP(<-0)     Y' = (Q - X3) B          V = S1 PPP(<-0)
//                                                         Y3 = Y' - V(<-3)
//   store                          X.zzz = ZZZ3           X.x = X3, X.y = Y3       X.zz = ZZ3
//
// (<-k) = value fetched from lane k of the quad. Per lane: two operands, one product and at most five kept values - about a
// third of the registers of the one-lane addition, so the wide element types run these kernels without scratch.
// Same formulas, operand classes and special cases as add_u / add_g and double_u / double_g (gmsm_curveu.h): the rare
// same-x case (P + P, P - P) is detected from PP: opposite points give infinity, equal points the quad doubling of X.
// A step of a kernel is: quad_load (all LDS reads) - barrier - compute + store - barrier, so that a record may be one
// quad's destination and another quad's source in the same step (in-place scans and trees).
#pragma once
#include "gmsm_curveu.h"

namespace gmsm {

#if defined(__HIPCC__)

// Orders a wave's own LDS accesses where the lanes of a quad exchange coordinates through a record that no other wave
// touches (stores of one lane before loads of another): the hardware executes a wave's LDS instructions in order, this
// keeps the compiler from reordering them.
__device__ __forceinline__ void quad_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <class U>
struct QRec {  // one extended-Jacobian value in LDS: x, y, zz, zzz + infinity flag
    U c[4];
    uint32_t inf;
    uint32_t pad[3];
};

// ---- the field operations of the quad formulas, per element class
template <class U> struct QF;
template <class P>
struct QF<FpU<P>> {  // bound-tracked prime field (bounds in multiples of q next to each use)
    using F = FpU<P>;
    template <bool INL> __device__ static __forceinline__ F mul(const F &a, const F &b) { return fmul<INL>(a, b); }
    __device__ static __forceinline__ F sub4(const F &a, const F &b) { return fpu_sub<P, 4>(a, b); }    // b < 4q
    __device__ static __forceinline__ F sub16(const F &a, const F &b) { return fpu_sub<P, 16>(a, b); }  // b < 16q
    __device__ static __forceinline__ F dbl(const F &a) { return fpu_dbl(a); }
    __device__ static __forceinline__ F triple(const F &a) { return fpu_add(fpu_add(a, a), a); }
    __device__ static __forceinline__ bool prod_is_zero(const F &a) { return fpu_prod_is_zero(a); }
    __device__ static __forceinline__ F shfl(const F &v, int src) {
        F o;
#pragma unroll
        for (int i = 0; i < P::UL; ++i) o.l[i] = (uint32_t)__shfl((int)v.l[i], src, 64);
        return o;
    }
    __device__ static __forceinline__ F select(bool first, const F &a, const F &b) {
        F o;
#pragma unroll
        for (int i = 0; i < P::UL; ++i) o.l[i] = first ? a.l[i] : b.l[i];
        return o;
    }
};
template <class P>
struct QF<Fp2U<P>> {  // reduced class R = [0, 4q): exact, no bound tracking
    using F = Fp2U<P>;
    using B = QF<FpU<P>>;
    template <bool INL> __device__ static __forceinline__ F mul(const F &a, const F &b) { return lz_mul<INL>(a, b); }
    __device__ static __forceinline__ F sub4(const F &a, const F &b) { return lz_sub(a, b); }
    __device__ static __forceinline__ F sub16(const F &a, const F &b) { return lz_sub(a, b); }
    __device__ static __forceinline__ F dbl(const F &a) { return lz_dbl(a); }
    __device__ static __forceinline__ F triple(const F &a) { return lz_add(lz_dbl(a), a); }
    __device__ static __forceinline__ bool prod_is_zero(const F &a) { return lz_is_zero(a); }
    __device__ static __forceinline__ F shfl(const F &v, int src) { return F{B::shfl(v.a0, src), B::shfl(v.a1, src)}; }
    __device__ static __forceinline__ F select(bool first, const F &a, const F &b) {
        return F{B::select(first, a.a0, b.a0), B::select(first, a.a1, b.a1)};
    }
};

// What a lane reads from LDS for one addition X += Y (before the barrier that separates reads from writes).
template <class U>
struct QAddOps {
    U a1, b1;   // level-1 operands of this lane
    U a2, b2;   // level-2 operands of the odd lanes (T2, T1); unused on the even ones
    bool xinf, yinf;
};

template <class U>
__device__ __forceinline__ QAddOps<U> quad_add_load(const QRec<U> *X, const QRec<U> *Y, uint32_t lane) {
    const uint32_t r = lane & 3u;
    QAddOps<U> o;
    o.xinf = X->inf != 0;
    o.yinf = Y->inf != 0;
    const QRec<U> *ra = (r & 1u) ? X : Y, *rb = (r & 1u) ? Y : X;
    o.a1 = ra->c[r >> 1];          // Y.x | X.x | Y.y | X.y
    o.b1 = rb->c[2 + (r >> 1)];    // X.zz | Y.zz | X.zzz | Y.zzz
    const uint32_t k2 = r == 1u ? 3u : 2u;  // lane 1: zzz (T2), lane 3: zz (T1)
    o.a2 = X->c[k2];
    o.b2 = Y->c[k2];
    return o;
}

// X = 2 X in place on a quad (dbl-2008-s-1, a = 0; g1.go:795-817); X not infinity. Only the quad's own record is touched:
// no other quad needs a barrier because of it, but the quad's own lanes read coordinates that other lanes of the quad
// wrote, so a barrier (or the end of the kernel's step) must separate it from the previous and the next access to X.
//   level   lane 0              lane 1                lane 2               lane 3
//   1       V = U U (U = 2y)    XX = x x              V = U U              XX = x x        (M = 3 XX)
//   2       W = U V             S = x V(<-0)          ZZ3 = V zz           MM = M M
//                                                                          X3 = MM - 2 S(<-1)
//   3       Wy = W y            ZZZ3 = W(<-0) zzz     -                    Y' = (S(<-1) - X3) M
//                                                                          Y3 = Y' - Wy(<-0)
template <class U, bool INL>
__device__ __forceinline__ void quad_dbl_inplace(QRec<U> *X, uint32_t lane) {
    using Q = QF<U>;
    const uint32_t r = lane & 3u;
    const int base = (int)(lane & ~3u);
    const bool odd = (r & 1u) != 0;
    const U c1 = X->c[odd ? 0 : 1];                        // x on the odd lanes, y on the even ones
    const U c2 = X->c[r == 1u ? 3 : 2];                    // zzz (lane 1), zz (lane 2); unused elsewhere
    const U a1 = Q::select(odd, c1, Q::dbl(c1));          // x | U = 2y < 14
    const U p1 = Q::template mul<INL>(a1, a1);            // V < 3 | XX < 2
    const U Vn = Q::shfl(p1, base);                        // V for lane 1
    const U M = Q::triple(p1);                             // odd lanes: 3 XX < 6
    // level 2: lane 0 (U, V), lane 1 (x, V), lane 2 (V, zz), lane 3 (M, M)
    const U a2 = Q::select(r == 3u, M, Q::select(r == 2u, p1, a1));
    const U b2 = Q::select(r == 3u, M, Q::select(r == 2u, c2, Q::select(r == 1u, Vn, p1)));
    const U p2 = Q::template mul<INL>(a2, b2);            // W, S, ZZ3, MM < 2
    const U W = Q::shfl(p2, base), S = Q::shfl(p2, base + 1);
    const U X3 = Q::sub4(p2, Q::dbl(S));                   // lane 3: MM - 2S < 6
    // level 3: lane 0 (W, y), lane 1 (W, zzz), lane 3 (S - X3, M)
    const U a3 = Q::select(r == 3u, Q::sub16(S, X3), W);
    const U b3 = Q::select(r == 3u, M, Q::select(r == 1u, c2, c1));
    const U p3 = Q::template mul<INL>(a3, b3);            // Wy, ZZZ3, (unused), Y' < 2
    const U Wy = Q::shfl(p3, base);
    if (r == 3u) {
        X->c[0] = X3;
        X->c[1] = Q::sub4(p3, Wy);                         // < 6
    } else if (r == 2u) {
        X->c[2] = p2;
    } else if (r == 1u) {
        X->c[3] = p3;
    }
}

// X += Y with the operands fetched by quad_add_load; call between two barriers. `active` = this quad has an addition
// to perform in this step (uniform inside the quad). A wave none of whose quads is active skips the products; otherwise
// every lane of the wave executes them (the exchanges are wave-wide instructions) and inactive quads do not store.
template <class U, bool INL>
__device__ __forceinline__ void quad_add_store(QRec<U> *X, const QAddOps<U> &o, bool active, uint32_t lane) {
    using Q = QF<U>;
    const uint32_t r = lane & 3u;
    const int base = (int)(lane & ~3u);
    const bool odd = (r & 1u) != 0;
    const bool compute = active && !o.xinf && !o.yinf;
    if (active && o.xinf && !o.yinf) {  // infinity + Y = Y: the coordinates of Y are among the operands (Y.x lane 0, Y.y lane 2,
        if (r == 0u) {                   // Y.zz and Y.zzz lane 1)
            X->c[0] = o.a1;
            X->inf = 0u;
        } else if (r == 2u) {
            X->c[1] = o.a1;
        } else if (r == 1u) {
            X->c[2] = o.b1;
            X->c[3] = o.b2;
        }
    }
    if (__ballot(compute) == 0ull) return;  // wave-uniform
    // level 1                                                                            prime-field bounds
    const U p1 = Q::template mul<INL>(o.a1, o.b1);                                       // U2, U1, S2, S1 < 2
    const U nb = Q::shfl(p1, (int)(lane ^ 1u));
    const U k1 = Q::select(odd, p1, Q::sub4(p1, nb));                                    // A < 6, U1, B < 6, S1
    // level 2
    const U p2 = Q::template mul<INL>(Q::select(odd, o.a2, k1), Q::select(odd, o.b2, k1));  // PP, T2, BB, T1 < 2
    const U PP = Q::shfl(p2, base);
    // level 3
    const U p3 = Q::template mul<INL>(Q::select(r == 3u, p2, k1), PP);                   // PPP, Q, (unused), ZZ3 < 2
    const U PPP = Q::shfl(p3, base), Qv = Q::shfl(p3, base + 1);
    const U X3 = Q::sub4(Q::sub4(p2, PPP), Q::dbl(Qv));                                   // lane 2: BB - PPP - 2Q < 10
    // level 4
    const U a4 = Q::select(r == 2u, Q::sub16(Qv, X3), Q::select(r == 1u, p2, k1));        // Q - X3 < 18 | T2 | S1
    const U p4 = Q::template mul<INL>(a4, Q::select(r == 2u, k1, PPP));                  // (unused), ZZZ3, Y', V < 2
    const U V = Q::shfl(p4, (int)(lane ^ 1u));
    if (!compute) return;
    if (Q::prod_is_zero(PP)) {  // same x (A == 0 <=> A^2 == 0): P + P or P - P, g1.go:757-765. Rare.
        const U BB = Q::shfl(p2, base + 2);
        if (Q::prod_is_zero(BB)) quad_dbl_inplace<U, INL>(X, lane);  // same point: 2 X (X's own record only)
        else if (r == 0u) X->inf = 1u;                                  // opposite points
        return;
    }
    if (r == 2u) {
        X->c[0] = X3;
        X->c[1] = Q::sub4(p4, V);  // Y' - V < 6
    } else if (r == 3u) {
        X->c[2] = p3;
    } else if (r == 1u) {
        X->c[3] = p4;
    }
}

// records in HBM (lazy XYZZ, infinity <=> zz limbs all zero) <-> LDS records, one coordinate per lane of the quad
// FRESH: a record k_accumulate_seg wrote (lz_rec_fresh, gmsm_curveu.h)
template <class U, bool FRESH = false>
__device__ __forceinline__ void quad_rec_load(QRec<U> *dst, const void *base, size_t index, bool present, uint32_t lane) {
    const uint32_t r = lane & 3u;
    const U *src = reinterpret_cast<const U *>(reinterpret_cast<const char *>(base) + index * sizeof(XYZZL<U>));
    bool inf = true;
    if (present) {
        const U zz = src[2];
        inf = lz_limbs_all_zero(zz);
        U c = src[r];
        if constexpr (FRESH) {
            if (!inf) lz_coord_fresh(c, r);
        }
        dst->c[r] = c;
    }
    if (r == 0u) dst->inf = inf ? 1u : 0u;
}
template <class U>
__device__ __forceinline__ void quad_rec_store(void *base, size_t index, const QRec<U> *src, uint32_t lane) {
    const uint32_t r = lane & 3u;
    U *dst = reinterpret_cast<U *>(reinterpret_cast<char *>(base) + index * sizeof(XYZZL<U>));
    U v = src->c[r];
    if (r == 2u && src->inf) v = lz_zero((const U *)nullptr);  // infinity <=> zz = 0
    dst[r] = v;
}

// ------------------------------------------------------------------ level 2 of the bucket reduction on quads
// grid = nwin_local, block = 4 * active threads (active = power of two >= nblocks1, 2 <= active <= 64). Quad j holds
// level-1 block j: (S_j, W_j), S_j already multiplied by the span when level 1 prescaled it.
// window_total = sum_j W_j + 2^log2span * sum_{j>=1} Suf_j, Suf = suffix sums of S: suffix scan (log2 active steps), quad
// doublings only if log2span != 0, one step W_j + Suf_j, tree (log2 active steps).
// dynamic LDS = 2 * active * sizeof(QRec<U>).
template <class U, bool INL>
__global__ void __launch_bounds__(256) k_reduce2_q(const void *__restrict__ in1, uint32_t nblocks1, uint32_t log2span,
                                                   uint32_t active, void *__restrict__ window_totals) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *S = reinterpret_cast<QRec<U> *>(lds_raw), *W = S + active;
    const uint32_t k = blockIdx.x, t = threadIdx.x, j = t >> 2, lane = t & 63u;
    quad_rec_load<U>(&S[j], in1, ((size_t)k * nblocks1 + j) * 2 + 0, j < nblocks1, lane);
    quad_rec_load<U>(&W[j], in1, ((size_t)k * nblocks1 + j) * 2 + 1, j < nblocks1, lane);
    __syncthreads();
    // inclusive suffix scan of S over the quads
#pragma nounroll
    for (uint32_t d = 1; d < active; d <<= 1) {
        const bool act = j + d < active;
        const QAddOps<U> o = quad_add_load<U>(&S[j], &S[act ? j + d : j], lane);
        __syncthreads();
        quad_add_store<U, INL>(&S[j], o, act, lane);
        __syncthreads();
    }
    // U-part of quad j: Suf_j for j >= 1, scaled by what is left of the span
    if (j == 0 && (t & 3u) == 0) S[0].inf = 1u;
    __syncthreads();
#pragma nounroll
    for (uint32_t s = 0; s < log2span; ++s) {
        if (!S[j].inf) quad_dbl_inplace<U, INL>(&S[j], lane);
        __syncthreads();  // the lanes of a quad exchange coordinates through the record: stores before the next loads
    }
    {
        const QAddOps<U> o = quad_add_load<U>(&W[j], &S[j], lane);
        quad_add_store<U, INL>(&W[j], o, true, lane);  // both records belong to this quad
    }
    __syncthreads();
    // tree over the quads
#pragma nounroll
    for (uint32_t d = active >> 1; d >= 1; d >>= 1) {
        const bool act = j < d;
        const QAddOps<U> o = quad_add_load<U>(&W[j], &W[act ? j + d : j], lane);
        __syncthreads();
        quad_add_store<U, INL>(&W[j], o, act, lane);
        __syncthreads();
    }
    if (t < 4) {  // quad 0 converts the total: canonical saturated XYZZ for the host
        using T = LzTraits<U>;
        using Mem = XYZZ<typename T::Sat>;
        typename T::Sat *dst = reinterpret_cast<typename T::Sat *>(reinterpret_cast<char *>(window_totals) + (size_t)k * sizeof(Mem));
        typename T::Sat v = T::template to_sat<INL>(W[0].c[t]);
        if (W[0].inf) {  // (1, 1, 0, 0) like g1JacExtended.SetInfinity (g1.go:688)
            const Mem inf = Mem::infinity();
            v = t == 0 ? inf.x : t == 1 ? inf.y : t == 2 ? inf.zz : inf.zzz;
        }
        dst[t] = v;
    }
}

// ------------------------------------------------------------------ level-1 combine on quads (split reduction)
// k_reduce_serial has left (S_g, W_g) of thread g = L consecutive buckets; this kernel combines N consecutive pairs into the
// block's (S_blk, W_blk): grid = (nblocks1, nwin), block = 4 N threads, quad t holds pair t of the block.
//   W_blk = sum_t W_t + L * sum_{t>=1} Suf_t,  Suf = inclusive suffix sums of S;  S_blk = Suf_0, stored as 2^prescale S_blk.
// Steps: suffix scan of S (log2 N; in place), then two trees at once - quads [0, N/2) over Suf_1.. (slot 0 reads as
// infinity in the first step), quads [N/2, N) over W -, then quad 0 doubles U log2L times and adds it to W.
// Quad N-1 belongs to the W tree in its first step only; from the second tree step on it doubles the parked S_blk once
// per step: log2 N - 1 tree steps and log2L + 1 finishing steps remain - exactly the log2span = log2L + log2 N doublings
// of the prescale.
// dynamic LDS = (2 N + 1) * sizeof(QRec<U>).
template <class U, bool INL, int N>
__global__ void __launch_bounds__(4 * N) k_combine_q(uint32_t log2L, void *__restrict__ out1, uint32_t prescale,
                                                     const void *__restrict__ pre, uint32_t T) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *S = reinterpret_cast<QRec<U> *>(lds_raw), *W = S + N, *PARK = S + 2 * N;
    const uint32_t k = blockIdx.y, blk = blockIdx.x, t = threadIdx.x, j = t >> 2, lane = t & 63u;
    const uint32_t g = blk * N + j;
    quad_rec_load<U>(&S[j], pre, ((size_t)k * T + g) * 2 + 0, g < T, lane);
    quad_rec_load<U>(&W[j], pre, ((size_t)k * T + g) * 2 + 1, g < T, lane);
    __syncthreads();
    constexpr uint32_t LG = N == 64 ? 6 : N == 32 ? 5 : N == 16 ? 4 : N == 128 ? 7 : 0;
    static_assert(LG != 0, "N must be 16, 32, 64 or 128");
#pragma nounroll
    for (uint32_t s = 0; s < LG; ++s) {  // suffix scan
        const uint32_t d = 1u << s;
        const bool act = j + d < (uint32_t)N;
        const QAddOps<U> o = quad_add_load<U>(&S[j], &S[act ? j + d : j], lane);
        __syncthreads();
        quad_add_store<U, INL>(&S[j], o, act, lane);
        __syncthreads();
    }
    if (j == 0) {  // park S_blk, then slot 0 counts as infinity for the U tree
        PARK->c[t & 3u] = S[0].c[t & 3u];
        if ((t & 3u) == 0) {
            PARK->inf = S[0].inf;
            S[0].inf = 1u;
        }
    }
    __syncthreads();
    const bool upper = j >= (uint32_t)N / 2;
    const uint32_t jj = upper ? j - N / 2 : j;
    QRec<U> *arr = upper ? W : S;
    uint32_t dbl_left = prescale;
    const uint32_t n_tree = LG, n_fin = log2L + 1;
#pragma nounroll
    for (uint32_t s = 0; s < n_tree + n_fin; ++s) {
        bool act = false, fin_dbl = false;
        QRec<U> *X = &arr[jj], *Y = &arr[jj];
        if (s < n_tree) {
            const uint32_t d = (uint32_t)N >> (s + 1);  // the tree halves: slots [0, d) += slots [d, 2d)
            if (d >= 1 && jj < d) {
                act = true;
                Y = &arr[jj + d];
            }
        } else if (s < n_tree + n_fin) {
            const uint32_t step = s - n_tree;
            if (j == 0) {
                if (step < log2L) fin_dbl = true;  // U <- 2 U
                else {                               // W <- W + U
                    act = true;
                    X = &W[0];
                    Y = &S[0];
                }
            }
        }
        // the doubler: quad N-1 is free from the second tree step on (d < N/2 => jj = N/2 - 1 >= d)
        const bool park_dbl = j == (uint32_t)N - 1 && s >= 1 && dbl_left > 0;
        const QAddOps<U> o = quad_add_load<U>(X, Y, lane);
        __syncthreads();
        quad_add_store<U, INL>(X, o, act, lane);
        if (fin_dbl && !S[0].inf) quad_dbl_inplace<U, INL>(&S[0], lane);
        if (park_dbl) {
            if (!PARK->inf) quad_dbl_inplace<U, INL>(PARK, lane);
        }
        if (s >= 1 && dbl_left > 0) --dbl_left;
        __syncthreads();
    }
    if (j == 0) quad_rec_store<U>(out1, ((size_t)k * gridDim.x + blk) * 2 + 1, &W[0], lane);
    if (j == 1) quad_rec_store<U>(out1, ((size_t)k * gridDim.x + blk) * 2 + 0, PARK, lane);
}

// ------------------------------------------------------------------ work-efficient level-1 combine (N = 64)
// k_combine_q computes all 64 suffix sums (Hillis-Steele: 6 steps of 64 additions) only to add them up again. What the
// block needs is A = sum_t S_t, U = sum_t t S_t and sum_t W_t, and U = sum_l 2^l M_l with M_l = the sum of the S_t whose
// index has bit l set. Steps 1..6 form, in place, the pair sums per index bit - P^0 = S, P^(l+1)_j = P^l_2j + P^l_(2j+1),
// P^l_j living in slot j << l - which leave the odd elements P^l_(2j+1) untouched in their slots t = (2j+1) << l; from the
// step after a level has read them, those slots are summed by a halving tree (M_l ends in slot 1 << l); the W tree runs
// alongside. Every step has (s + 1) groups of g = 32 >> (s-1) additions: 64, 48, 32, 20, 12, 7 - 3.4 additions per pair
// instead of 8, and waves without work skip the products. The tail combines the six M_l:
//   U = (M_0 + 2 M_1) + 4 (M_2 + 2 M_3) + 16 (M_4 + 2 M_5), then L U, then W + L U      (7 + log2L + 1 steps on quads 0..2)
// while quad 15 doubles the parked A log2span times. 17 steps at L = 8 (k_combine_q: 16), about half the issue work: for
// the element types whose combine is throughput-bound (several workgroups per CU: the 9- and 14-limb fields).
// tests/test_combine_model.py runs this step machine over Z. Launch and LDS as k_combine_q<U, INL, 64>.
template <class U, bool INL>
__global__ void __launch_bounds__(256) k_combine_we(uint32_t log2L, void *__restrict__ out1, uint32_t prescale,
                                                    const void *__restrict__ pre, uint32_t T) {
    constexpr uint32_t N = 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    QRec<U> *S = reinterpret_cast<QRec<U> *>(lds_raw), *W = S + N, *PARK = S + 2 * N;
    const uint32_t k = blockIdx.y, blk = blockIdx.x, t = threadIdx.x, j = t >> 2, lane = t & 63u;
    const uint32_t g0 = blk * N + j;
    quad_rec_load<U>(&S[j], pre, ((size_t)k * T + g0) * 2 + 0, g0 < T, lane);
    quad_rec_load<U>(&W[j], pre, ((size_t)k * T + g0) * 2 + 1, g0 < T, lane);
    __syncthreads();
#pragma nounroll
    for (uint32_t s = 1; s <= 6; ++s) {
        const uint32_t g = 32u >> (s - 1), G = j / g, i = j % g;
        const bool act = G <= s;
        QRec<U> *X = &S[0], *Y = &S[0];
        if (G == 0) {  // pair sums of level l = s - 1
            const uint32_t l = s - 1;
            X = &S[(2 * i) << l];
            Y = &S[(2 * i + 1) << l];
        } else if (G < s) {  // tree over the odd elements of level l = G - 1: slot i += slot i + g
            const uint32_t l = G - 1;
            X = &S[(2 * i + 1) << l];
            Y = &S[(2 * (i + g) + 1) << l];
        } else if (G == s) {  // W tree
            X = &W[i];
            Y = &W[i + g];
        }
        const QAddOps<U> o = quad_add_load<U>(X, Y, lane);
        __syncthreads();
        quad_add_store<U, INL>(X, o, act, lane);
        __syncthreads();
    }
    if (j == 0) {  // park A = S[0] for the prescaling doubler
        PARK->c[t & 3u] = S[0].c[t & 3u];
        if ((t & 3u) == 0) PARK->inf = S[0].inf;
    }
    __syncthreads();
    uint32_t dbl_left = prescale;
    const uint32_t n_tail = 7 + log2L + 1;
#pragma nounroll
    for (uint32_t s = 0; s < n_tail; ++s) {
        // tail program of quads 0, 1, 2 on the slots of M_0..M_5 = S[1], S[2], S[4], S[8], S[16], S[32]
        int op = 0;  // 0 nothing, 1 double X, 2 X += Y
        QRec<U> *X = &S[0], *Y = &S[0];
        if (j < 3) {
            if (s == 0) { op = 1; X = &S[2u << (2 * j)]; }                                    // 2 M_1, 2 M_3, 2 M_5
            else if (s == 1) { op = 2; X = &S[1u << (2 * j)]; Y = &S[2u << (2 * j)]; }       // M_0 + 2 M_1, M_2 + 2 M_3, M_4 + 2 M_5
            else if (s == 2 || s == 3) { if (j >= 1) { op = 1; X = &S[j == 1 ? 4 : 16]; } }   // x4 under way, x16 under way
            else if (s == 4) { if (j == 0) { op = 2; X = &S[1]; Y = &S[4]; } else if (j == 2) { op = 1; X = &S[16]; } }
            else if (s == 5) { if (j == 2) { op = 1; X = &S[16]; } }
            else if (s == 6) { if (j == 0) { op = 2; X = &S[1]; Y = &S[16]; } }             // U complete
            else if (s < 7 + log2L) { if (j == 0) { op = 1; X = &S[1]; } }                    // L U
            else if (j == 0) { op = 2; X = &W[0]; Y = &S[1]; }                                // W + L U
        }
        const bool park_dbl = j == 15 && dbl_left > 0;
        const QAddOps<U> o = quad_add_load<U>(X, Y, lane);
        __syncthreads();
        quad_add_store<U, INL>(X, o, op == 2, lane);
        if (op == 1 && !X->inf) quad_dbl_inplace<U, INL>(X, lane);
        if (park_dbl && !PARK->inf) quad_dbl_inplace<U, INL>(PARK, lane);
        if (dbl_left > 0) --dbl_left;
        __syncthreads();
    }
    if (j == 0) quad_rec_store<U>(out1, ((size_t)k * gridDim.x + blk) * 2 + 1, &W[0], lane);
    if (j == 1) quad_rec_store<U>(out1, ((size_t)k * gridDim.x + blk) * 2 + 0, PARK, lane);
}

#endif  // __HIPCC__

}  // namespace gmsm
