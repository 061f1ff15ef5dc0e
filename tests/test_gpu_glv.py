"""GLV half scalars in the sorted pipeline (GMSM_OPT_GLV = 2; gmsm_glv.h: k_convert_points_glv writes P_i and phi(P_i),
k_decompose_glv the digits of k1_i, k2_i; half the windows, the same additions): the affine result is the reference's, limb
for limb - ecc.SplitScalar / mulGLV (ecc/utils.go:141-170, ecc/bn254/g1.go:536-600) change how a multiple is computed, not
which point it is."""
import numpy as np
import pytest

from conftest import ALL_GROUPS, random_scalars, rng_for, scalars_from_ints

pytestmark = pytest.mark.gpu


def _inputs(o, g, n, seed):
    rng = rng_for(43, g.gid, n, seed)
    pts = o.gen_points(n, 4321 + n, 99, nthreads=8)
    sc = random_scalars(rng, g.curve, n)
    pts[[5, 17]] = 0
    pts[11] = pts[3]
    sc[11] = sc[3]
    sc[20] = 0
    sc[21:27] = scalars_from_ints(g.curve, [1, 2, g.curve.r - 1, 1 << 64, (1 << 128) + 5, g.curve.lambda_glv])
    return pts, sc


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_glv_pipeline_matches_oracle(gm, oracle_mod, curve, which):
    """Device entry and host entry (point ranges share one bucket set), the library's width and forced widths, against the
    oracle; the same call with GLV off gives the same limbs."""
    import torch
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    gj = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    o = oracle_mod.Oracle(curve, which)
    n = 20011
    pts, sc = _inputs(o, g, n, 0)
    expected = o.msm_affine(pts, sc, nthreads=8)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    with gm.options(glv=2, small_bits=1):
        aff, err = g.MultiExp(pts, sc)
        assert err is None and (aff == expected).all()
        assert (gj.jac_to_affine(gj.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)) == expected).all()
        for c in (10, 11, 13, 16, 17):
            with gm.options(window_bits=c):
                assert (gj.jac_to_affine(gj.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)) == expected).all(), c
        with gm.options(host_ranges=3):
            aff, err = g.MultiExp(pts, sc)
            assert err is None and (aff == expected).all()
        with gm.options(max_run=6000):  # device-side point ranges: merged buckets, one reduction
            assert (gj.jac_to_affine(gj.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)) == expected).all()
        m = 700  # a prefix (the sorted pipeline at a small size)
        assert (gj.jac_to_affine(gj.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), m)) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all()
    with gm.options(glv=0, small_bits=1):
        assert (gj.jac_to_affine(gj.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)) == expected).all()


@pytest.mark.parametrize("kind", ["all_equal", "smallvalues", "edge_scalars"])
def test_glv_pipeline_skewed_scalars(gm, oracle_mod, kind):
    """Crowded buckets under GLV (BN254 G1, 2^16 points): all scalars equal, the reference's 'smallvalues' distribution
    (multiexp_test.go:319-325), and the edge values of the split (0-adjacent, r-adjacent, lambda, the lattice's own coordinates)."""
    import torch
    import importlib
    curves = importlib.import_module("gnark-crypto_amd.curves")
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = 1 << 16
    rng = rng_for(44, n)
    pts = o.gen_points(n, 77, 5, nthreads=8)
    sc = random_scalars(rng, g.curve, n)
    if kind == "all_equal":
        sc = np.ascontiguousarray(np.tile(sc[:1], (n, 1)))
    elif kind == "smallvalues":
        sc[::5] = 0
        sc[::5, 0] = 1
    else:
        glv = curves.GlvParams(g.curve)
        r, lam = g.curve.r, g.curve.lambda_glv
        vals = [1, 2, r - 1, r - 2, lam, lam + 1, r - lam, lam - 1, r // 2, (r + 1) // 2, 1 << 127, (1 << 128) - 1, 1 << 253]
        vals += [v for k in range(1, 6) for v in glv.a if False] + [abs(x) % r for x in glv.a]  # the lattice's own coordinates
        assert all((sum(glv.split(v)[i] * (1, lam)[i] for i in range(2)) - v) % r == 0 for v in vals)
        vals = (vals * 64)[:64]
        sc[:64] = scalars_from_ints(g.curve, vals)
    expected = o.msm_affine(pts, sc, nthreads=8)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    with gm.options(glv=2):
        assert (g.jac_to_affine(g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)) == expected).all()


@pytest.mark.parametrize("curve,which", [("bls12_381", "g1"), ("bn254", "g2"), ("bw6_761", "g1")])
def test_glv_off_is_the_integer_combination_outside_the_subgroup(gm, oracle_mod, pyref_mod, curve, which):
    """The precondition include/gmsm.h states for GMSM_OPT_GLV: phi(P) = [lambda]P holds on the r-torsion only (the reference's
    mulGLV has the same one, ecc/bn254/g1.go:536-600), while the reference's MultiExp - which never uses the endomorphism - is
    the integer combination sum s_i P_i on ANY point of the curve. With GLV off the engine is exactly that on points with a
    cofactor component too (fused kernel and sorted pipeline, against the oracle, which is plain bucket arithmetic); with GLV on
    the call still succeeds, and agrees once every point is in the subgroup (the other tests of this file)."""
    from subgroup_points import curve_points, times_r
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle_mod.Oracle(curve, which)
    pg = pyref_mod.Group(g.curve, which)
    n = 300
    pts = o.gen_points(n, 77, 5, nthreads=4)
    outside = [P for P in curve_points(pyref_mod, pg, 40, start=2) if times_r(pg, P) is not None][:8]
    assert len(outside) >= 4
    for k, P in enumerate(outside):
        pts[7 + 31 * k] = np.array(pg.point_to_limbs(P), dtype=np.uint64)
    sc = random_scalars(rng_for(47, g.gid, n), g.curve, n)
    expected = o.msm_affine(pts, sc, nthreads=4)
    # the oracle's sum really has a component outside the subgroup (otherwise this test shows nothing)
    assert times_r(pg, pg.point_from_limbs(expected)) is not None
    with gm.options(glv=0):
        aff, err = g.MultiExp(pts, sc)                       # the fused small-n kernel
        assert err is None and (aff == expected).all()
        with gm.options(small_bits=1):
            aff, err = g.MultiExp(pts, sc)                   # the sorted pipeline
            assert err is None and (aff == expected).all()
    aff, err = g.MultiExp(pts, sc)                           # default (GLV on): defined, but a different combination
    assert err is None
