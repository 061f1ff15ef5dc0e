"""GPU parity tests of the fused small-n kernel (gnark-crypto_amd/csrc/gmsm_small.h: one launch, one workgroup per window and
slice, buckets in LDS) against the CPU oracle: the sizes the reference benches from (multiexp_test.go:344) and Pedersen /
KZG batch verification issue (fr/pedersen/pedersen.go:100-131)."""
import numpy as np
import pytest

from conftest import ALL_GROUPS, random_scalars, rng_for, scalars_from_ints

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 31, 73, 257, 1023, 4099]


def _small_runs(gm):
    return int(gm._lib.load().gmsm_debug_small_runs())


def _inputs(o, g, n, seed):
    rng = rng_for(41, g.gid, n, seed)
    pts = o.gen_points(n, 1234 + n, 77, nthreads=8)
    sc = random_scalars(rng, g.curve, n)
    if n >= 31:  # infinities, duplicated (point, scalar) pairs, zero scalars, edge values (SURVEY.md 8(d) adversarial set)
        pts[[5, 17]] = 0
        pts[11] = pts[3]
        sc[11] = sc[3]
        pts[13] = pts[7]
        sc[20] = 0
        sc[21:26] = scalars_from_ints(g.curve, [1, 2, g.curve.r - 1, 1 << 64, (1 << 128) + 5])
    return pts, sc


# the forms of the fused kernel: GLV half scalars on / off (GMSM_OPT_GLV) x bucket phase on lane quads never / always
# (GMSM_OPT_SMALL_QUAD; the Fp2 groups and BW6-761 are built with the quad form only, whatever the switch says)
FORMS = [(1, 0), (0, 1), (1, 2), (0, 2), (1, 1)]


@pytest.mark.parametrize("glv,quad", FORMS)
@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_small_kernel_matches_oracle(gm, oracle_mod, curve, which, glv, quad):
    """Every size through the fused kernel (one and several slices per window, one and several chunks per workgroup of the quad
    form), host entry and device entry, with and without GLV half scalars, against the oracle's MultiExp on the same input."""
    import torch
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle_mod.Oracle(curve, which)
    with gm.options(small_max=8192, glv=glv, small_quad=quad):
        for n in SIZES:
            pts, sc = _inputs(o, g, n, 0)
            expected = o.msm_affine(pts, sc, nthreads=8)
            before = _small_runs(gm)
            aff, err = g.MultiExp(pts, sc)
            assert err is None and (aff == expected).all(), n
            d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
            d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
            jac = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
            assert (g.jac_to_affine(jac) == expected).all(), n
            assert _small_runs(gm) == before + 2, n  # both calls took the fused kernel


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bls12_381", "g2"), ("bw6_761", "g1")])
def test_small_kernel_every_width_and_skew(gm, oracle_mod, curve, which):
    """Forced window widths 2..7 (the top window's short digit, the borrow chain), all scalars equal / all points equal
    (one run of maximal length: the doubling steps inside a run, the P + P and P - P cases of the addition), all-zero
    scalars and all points at infinity."""
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle_mod.Oracle(curve, which)
    n = 600
    pts, sc = _inputs(o, g, n, 1)
    expected = o.msm_affine(pts, sc, nthreads=8)
    for c in range(2, 8):
        for glv, quad in ((1, 0), (0, 1), (1, 2)):
            with gm.options(small_bits=c, glv=glv, small_quad=quad):
                before = _small_runs(gm)
                aff, err = g.MultiExp(pts, sc)
                assert err is None and (aff == expected).all(), (c, glv, quad)
                assert _small_runs(gm) == before + 1
    equal = np.tile(sc[:1], (n, 1))
    assert (g.MultiExp(pts, equal)[0] == o.msm_affine(pts, equal, nthreads=8)).all()
    same = np.tile(pts[:1], (n, 1))
    assert (g.MultiExp(same, equal)[0] == o.msm_affine(same, equal, nthreads=8)).all()
    assert (g.MultiExp(same, sc)[0] == o.msm_affine(same, sc, nthreads=8)).all()
    neg = same.copy()  # P and -P alternate with equal scalars: the running sums pass through infinity
    neg[1::2] = o.msm_affine(same[:1], scalars_from_ints(g.curve, [g.curve.r - 1]))
    assert (g.MultiExp(neg, equal)[0] == o.msm_affine(neg, equal, nthreads=8)).all()
    zeros = np.zeros_like(sc)
    assert (g.MultiExp(pts, zeros)[0] == 0).all()
    assert (g.MultiExp(np.zeros_like(pts), sc)[0] == 0).all()


def test_small_kernel_over_registered_bases_and_off_switch(gm, oracle_mod):
    """Registered bases (the rewritten form + infinity flags) through the fused kernel; GMSM_OPT_SMALL_BITS = 1 switches it
    off and the sorted pipeline returns the same point."""
    import torch
    g = gm.G1Affine("bn254")
    gj = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = 1500
    pts, sc = _inputs(o, g, n, 2)
    expected = o.msm_affine(pts, sc, nthreads=8)
    rb = gj.register_bases(points=pts)
    try:
        for m in (n, 700, 3):
            before = _small_runs(gm)
            jac, err = rb.MultiExp(sc[:m], gm.MultiExpConfig())
            assert err is None and (gj.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all(), m
            d_sc = torch.from_numpy(sc[:m].view(np.int64)).cuda()
            assert (gj.jac_to_affine(rb.multiexp_device(d_sc.data_ptr(), m)) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all(), m
            assert _small_runs(gm) == before + 2
    finally:
        rb.release()
    with gm.options(small_bits=1):
        before = _small_runs(gm)
        aff, err = g.MultiExp(pts, sc)
        assert err is None and (aff == expected).all()
        assert _small_runs(gm) == before


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_small_kernel_over_narrow_window_tables(gm, oracle_mod, curve, which):
    """gmsm_bases_precompute also builds narrow tables (2^(6 w) P_i for the first 4096 bases; 2048 for BW6-761): calls of a few thousand points
    over the handle then run the fused kernel with ONE bucket set per workgroup and no host-side fold. Every size (one slice,
    many slices, a slice that straddles two windows), infinities / duplicates / zero scalars, host and device scalars, a
    prefix longer than the narrow tables (falls back to the plain forms), and the tables switched off."""
    import torch
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    gj = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    o = oracle_mod.Oracle(curve, which)
    nreg = 5000
    pts, sc = _inputs(o, g, nreg, 3)
    rb = gj.register_bases(points=pts)
    table_runs = lambda: int(gm._lib.load().gmsm_debug_table_runs())
    try:
        rb.precompute(0)
        top = 2048 if curve == "bw6_761" else 4096  # the narrow tables' reach (Group::SMALL_TABLE_POINTS)
        for m in (1, 2, 5, 6, 31, 257, 1023, top):
            expected = o.msm_affine(pts[:m], sc[:m], nthreads=8)
            before, tb = _small_runs(gm), table_runs()
            jac, err = rb.MultiExp(sc[:m], gm.MultiExpConfig())
            assert err is None and (gj.jac_to_affine(jac) == expected).all(), m
            d_sc = torch.from_numpy(np.ascontiguousarray(sc[:m]).view(np.int64)).cuda()
            assert (gj.jac_to_affine(rb.multiexp_device(d_sc.data_ptr(), m)) == expected).all(), m
            assert _small_runs(gm) == before + 2 and table_runs() == tb + 2, m  # fused kernel, through the narrow tables
        # all scalars zero / equal: the total is infinity / one crowded bucket
        zeros = np.zeros_like(sc[:300])
        assert (gj.jac_to_affine(rb.MultiExp(zeros, gm.MultiExpConfig())[0]) == 0).all()
        equal = np.tile(sc[:1], (900, 1))
        assert (gj.jac_to_affine(rb.MultiExp(equal, gm.MultiExpConfig())[0]) == o.msm_affine(pts[:900], equal, nthreads=8)).all()
        # a prefix beyond the narrow tables, and the tables switched off: same points from the other forms
        m = 4500
        expected = o.msm_affine(pts[:m], sc[:m], nthreads=8)
        assert (gj.jac_to_affine(rb.MultiExp(sc[:m], gm.MultiExpConfig())[0]) == expected).all()
        with gm.options(tables=0):
            tb = table_runs()
            assert (gj.jac_to_affine(rb.MultiExp(sc[:700], gm.MultiExpConfig())[0]) == o.msm_affine(pts[:700], sc[:700], nthreads=8)).all()
            assert table_runs() == tb
    finally:
        rb.release()


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bw6_761"])
def test_glv_split_on_the_device(gm, curve):
    """gmsm_glv.h splits every scalar into half scalars k1 + k2 lambda = s mod r (ecc.SplitScalar, ecc/utils.go:141-170, with the
    lattice of PrecomputeLattice :62-135; lambdaGLV bn254.go:133): the congruence and the bound |k| < 2^GLV_BITS on the
    device's own output, for random scalars and the edge values, and word for word against the big-int statement of the same
    formulas (gnark-crypto_amd/curves.py GlvParams.split)."""
    import importlib
    curves = importlib.import_module("gnark-crypto_amd.curves")
    c = curves.CURVES[curve]
    glv = curves.GlvParams(c)
    g = gm.G1Jac(curve)
    rng = rng_for(97, g.gid)
    vals = [0, 1, 2, c.r - 1, c.r - 2, c.r // 2, c.lambda_glv, c.r - c.lambda_glv, 1 << (c.r.bit_length() - 1)]
    vals += [int.from_bytes(rng.bytes(64), "little") % c.r for _ in range(500)]
    sc = scalars_from_ints(c, vals)
    hl = glv.hl
    out = np.zeros((len(vals), 2, hl + 1), dtype=np.uint32)
    rc = gm._lib.load().gmsm_debug_glv_split(g.gid, sc.ctypes.data, len(vals), out.ctypes.data)
    assert rc == 0, gm._lib.last_error()
    for s, rec in zip(vals, out):
        ks = []
        for h in range(2):
            mag = sum(int(w) << (32 * i) for i, w in enumerate(rec[h, 1:]))
            assert mag < 1 << glv.bits
            ks.append(-mag if rec[h, 0] else mag)
        assert (ks[0] + ks[1] * c.lambda_glv - s) % c.r == 0, s
        assert tuple(ks) == glv.split(s), s
