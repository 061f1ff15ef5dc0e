"""The host-side window fold (gmsm_fold_windows / gmsm_fold_window_sets; msmReduceChunk, ecc/bn254/multiexp.go:302-315) needs
no GPU: Horner over the window totals, its doublings in Jacobian coordinates (Group::fold). Checked against the oracle:
fold(T_0..T_{nwin-1}) = sum_w [2^(c w)] T_w, with windows at infinity, equal neighbouring totals (P + P inside the
Horner step) and a total that cancels the running sum."""
import numpy as np
import pytest

from conftest import ALL_GROUPS, scalars_from_ints


def _one(g, which):
    L = g.coord_limbs
    base = (g.curve.p.bit_length() + 63) // 64  # limbs of the base field; L = 2 * base over Fp2 (BN254 / BLS12-381 G2)
    R = (1 << (64 * base)) % g.curve.p
    one = np.zeros(L, dtype=np.uint64)
    one[:base] = [(R >> (64 * i)) & (2**64 - 1) for i in range(base)]
    return one


def _totals(g, which, pts):
    """XYZZ records (x, y, 1, 1) of affine points; all-zero affine rows become infinity (1, 1, 0, 0)."""
    L = g.coord_limbs
    one = _one(g, which)
    tot = np.zeros((pts.shape[0], g.xyzz_limbs), dtype=np.uint64)
    for w in range(pts.shape[0]):
        if pts[w].any():
            tot[w, :2 * L] = pts[w]
            tot[w, 2 * L:3 * L] = one
            tot[w, 3 * L:] = one
        else:
            tot[w, :L] = one
            tot[w, L:2 * L] = one
    return tot


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_fold_windows_equals_weighted_sum(gm, oracle_mod, curve, which):
    g = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    o = oracle_mod.Oracle(curve, which)
    for c in (16, 13, 5):
        nwin = g.num_windows(c)
        pts = o.gen_points(nwin, 3 + c, 5, nthreads=1)
        pts[2] = 0                      # a window at infinity
        pts[nwin - 1] = 0               # the top window at infinity: the running sum starts empty
        pts[4] = pts[5]                 # equal totals
        weights = [0 if w in (2, nwin - 1) else (1 << (c * w)) % g.curve.r for w in range(nwin)]
        want = o.msm_affine(pts, scalars_from_ints(g.curve, weights), nthreads=2)
        jac = g.fold_windows(_totals(g, which, pts), c)
        assert (g.jac_to_affine(jac) == want).all(), c
    # all windows at infinity -> Z = 0
    c = 16
    nwin = g.num_windows(c)
    jac = g.fold_windows(_totals(g, which, np.zeros((nwin, g.aff_limbs), dtype=np.uint64)), c)
    assert not jac[2 * g.coord_limbs:].any()


def test_fold_running_sum_cancels(gm, oracle_mod):
    """T_0 = -[2^c] T_1: the Horner step adds a point to its own negative; the result is infinity."""
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    c, nwin = 16, g.num_windows(16)
    base = o.gen_points(1, 7, 1, nthreads=1)
    t1 = base[0]
    r = g.curve.r
    neg = o.msm_affine(base, scalars_from_ints(g.curve, [(r - (1 << c)) % r]), nthreads=1)   # -[2^c] T_1
    pts = np.zeros((nwin, g.aff_limbs), dtype=np.uint64)
    pts[1] = t1
    pts[0] = neg
    jac = g.fold_windows(_totals(g, "g1", pts), c)
    assert not jac[2 * g.coord_limbs:].any()


def test_fold_window_sets_adds_the_sets(gm, oracle_mod):
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    c, nwin = 16, g.num_windows(16)
    pts = o.gen_points(3 * nwin, 11, 3, nthreads=1)
    sets = np.stack([_totals(g, "g1", pts[s * nwin:(s + 1) * nwin]) for s in range(3)])
    weights = [(1 << (c * (i % nwin))) % g.curve.r for i in range(3 * nwin)]
    want = o.msm_affine(pts, scalars_from_ints(g.curve, weights), nthreads=2)
    assert (g.jac_to_affine(g.fold_window_sets(sets, c)) == want).all()
