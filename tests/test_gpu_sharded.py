"""One MultiExp over several devices INSIDE the library (gmsm_multiexp_sharded, gmsm_bases_register_sharded and the
drop-in entries once more than one device is configured), and the point-range splits of the single-device entries.

A GPU box of the test pool has ONE device, so the device list names device 0 several times: N logical ranks, each with
its own host thread, lease and pipeline run - the same code path that runs one rank per GPU on an 8-GPU node, minus the
second PCIe link.  Every result must equal the oracle bit for bit on the affine limbs.
Reference shape: one worker per window collected on a channel (ecc/bn254/multiexp.go:148-209), the split of the points
with AddAssign (:98-140), the fold (:302-315)."""
import threading

import numpy as np
import pytest

from conftest import random_scalars, rng_for, scalars_from_ints

pytestmark = pytest.mark.gpu


def _jac_group(gm, curve, which):
    return (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)


def _inputs(o, g, n, tag):
    rng = rng_for(31, g.gid, tag, n & 0xFFFFFF)
    pts = o.gen_points(n, int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62)), nthreads=16)
    sc = random_scalars(rng, g.curve, n)
    pts[[5, n // 2, n - 1]] = 0   # infinities in the first, a middle and the last slice
    sc[[7, n // 3]] = 0           # zero scalars
    sc[::11, 1:] = 0              # single-limb scalars
    return pts, sc


@pytest.mark.parametrize("curve,which,n", [("bn254", "g1", (1 << 19) + 5), ("bls12_381", "g1", (1 << 18) + 1),
                                           ("bn254", "g2", (1 << 17) + 3), ("bls12_381", "g2", (1 << 17) + 3),
                                           ("bw6_761", "g1", (1 << 17) + 9)])
def test_sharded_multiexp_logical_ranks(gm, oracle_mod, curve, which, n):
    """gmsm_multiexp_sharded with 2, 3 and 8 logical ranks on device 0, point and window decomposition."""
    g = _jac_group(gm, curve, which)
    o = oracle_mod.Oracle(curve, which)
    pts, sc = _inputs(o, g, n, 1)
    expected = o.msm_affine(pts, sc, nthreads=16)
    for ranks in (2, 3, 8):
        for mode in ("points", "windows"):
            if mode == "windows" and ranks == 3 and which == "g2":
                continue  # every window-mode rank copies all bases: keep the slow groups to two cases
            jac, err = g.MultiExpSharded(pts, sc, devices=[0] * ranks, mode=mode)
            assert err is None, err
            assert (g.jac_to_affine(jac) == expected).all(), (ranks, mode)
    # argument checks of the reference, same codes as the single-device entry
    _, err = g.MultiExpSharded(pts[:10], sc[:9], devices=[0, 0])
    assert err == "len(points) != len(scalars)"
    _, err = g.MultiExpSharded(pts[:10], sc[:10], gm.MultiExpConfig(NbTasks=1025), devices=[0, 0])
    assert err == "invalid config: config.NbTasks > 1024"
    jac, err = g.MultiExpSharded(pts[:0], sc[:0], devices=[0, 0])
    assert err is None and not jac[2 * g.coord_limbs:].any()  # Z = 0: infinity


def test_sharded_small_inputs_and_bad_devices(gm, oracle_mod):
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    for n in (1, 73, 4099, (1 << 16) + 1, (1 << 17) - 1):  # below two slices of 2^16: fewer ranks than asked for
        pts, sc = _inputs(o, g, max(n, 12), 2)
        pts, sc = pts[:n], sc[:n]
        for mode in ("points", "windows"):
            jac, err = g.MultiExpSharded(pts, sc, devices=[0] * 4, mode=mode)
            assert err is None, err
            assert (g.jac_to_affine(jac) == o.msm_affine(pts, sc, nthreads=4)).all(), (n, mode)
    ndev = gm._lib.load().gmsm_device_count()
    _, err = g.MultiExpSharded(pts, sc, devices=[0, ndev])
    assert err is not None and "does not exist" in err
    with pytest.raises(RuntimeError, match="does not exist"):
        gm.set_devices([ndev + 3])


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bls12_381", "g2")])
def test_sharded_registered_bases(gm, oracle_mod, curve, which):
    """gmsm_bases_register_sharded: full copies on every distinct device; MultiExp over prefixes through the ordinary
    gmsm_multiexp_bases entry and with the decomposition forced."""
    g = _jac_group(gm, curve, which)
    o = oracle_mod.Oracle(curve, which)
    n = (1 << 18) + 77 if which == "g1" else (1 << 17) + 77
    pts, sc = _inputs(o, g, n, 3)
    rb = g.register_bases_sharded(pts, devices=[0, 0, 0])
    try:
        for m in (n, (1 << 17) + 1, 3000, 1, 0):
            exp = o.msm_affine(pts[:m], sc[:m], nthreads=16)
            jac, err = rb.MultiExp(sc[:m])
            assert err is None, err
            assert (g.jac_to_affine(jac) == exp).all(), m
            for mode in ("points", "windows"):
                jac, err = rb.MultiExpSharded(sc[:m], mode=mode)
                assert err is None, err
                assert (g.jac_to_affine(jac) == exp).all(), (m, mode)
        _, err = rb.MultiExp(np.zeros((n + 1, g.fr_limbs), dtype=np.uint64))
        assert err == "len(points) != len(scalars)"
        _, err = rb.MultiExp(sc, gm.MultiExpConfig(NbTasks=2000))
        assert err == "invalid config: config.NbTasks > 1024"
        # k independent MultiExp over the sharded bases: block r of the vectors on rank r, no exchange (replica mode)
        m = 5000
        vecs = np.stack([random_scalars(rng_for(33, j), g.curve, m) for j in range(5)])
        jacs, err = rb.MultiExpBatch(scalars=vecs)
        assert err is None, err
        for j in range(5):
            assert (g.jac_to_affine(jacs[j]) == o.msm_affine(pts[:m], vecs[j], nthreads=8)).all(), j
        jacs, err = rb.MultiExpBatch(scalars=vecs[:2])  # fewer vectors than ranks
        assert err is None and (g.jac_to_affine(jacs[1]) == o.msm_affine(pts[:m], vecs[1], nthreads=8)).all()
    finally:
        rb.release()


def test_dropin_entries_shard_over_configured_devices(gm, oracle_mod):
    """gmsm_set_devices: the drop-in entries (gmsm_multiexp, _affine, gmsm_fold) spread the call themselves; calls from
    several threads at once share the rank workers."""
    g = gm.G1Jac("bn254")
    ga = gm.G1Affine("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = (1 << 18) + 3
    pts, sc = _inputs(o, g, n, 4)
    expected = o.msm_affine(pts, sc, nthreads=16)
    L = gm._lib.load()
    try:
        gm.set_devices([0, 0, 0])
        assert gm.get_devices() == [0, 0, 0]
        jac, err = g.MultiExp(pts, sc)
        assert err is None and (g.jac_to_affine(jac) == expected).all()
        aff, err = ga.MultiExp(pts, sc)
        assert err is None and (aff == expected).all()
        # Fold: sum_i points[i] coeff^i
        coeff = random_scalars(rng_for(32), g.curve, 1)[0]
        m = (1 << 17) + 1
        fr = oracle_mod.Field("bn254_fr", g.fr_limbs)
        gamma = sum(int(v) << (64 * k) for k, v in enumerate(fr.from_mont(coeff)))
        vals, acc = [], 1
        for _ in range(m):
            vals.append(acc)
            acc = acc * gamma % g.curve.r
        powers = scalars_from_ints(g.curve, vals)
        aff, err = ga.Fold(pts[:m], coeff)
        assert err is None and (aff == o.msm_affine(pts[:m], powers, nthreads=16)).all()
        # four callers at once
        res = [None] * 4

        def caller(i):
            res[i] = g.MultiExp(pts, sc)

        th = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for jac, err in res:
            assert err is None and (g.jac_to_affine(jac) == expected).all()
    finally:
        gm.set_devices(None)
    assert len(gm.get_devices()) == 1  # nothing configured: no spreading (opt-in), the calling thread's device


# ------------------------------------------------------------------ point-range splits of the single-device entries
@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bls12_381", "g2")])
def test_pipeline_run_cap_splits_every_entry(gm, oracle_mod, curve, which, forced_options):
    """Inputs beyond one pipeline run (2^27 points; GMSM_OPT_MAX_RUN lowers the cap) are cut into point ranges whose window
    totals are added - the reference's split + AddAssign (multiexp.go:98-140): device entry, registered bases (device and
    host scalars) and the host entry, n not a multiple of the run."""
    import torch
    g = _jac_group(gm, curve, which)
    o = oracle_mod.Oracle(curve, which)
    n = 3 * 4096 + 77
    pts, sc = _inputs(o, g, n, 5)
    expected = o.msm_affine(pts, sc, nthreads=8)
    forced_options(max_run=4096)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    assert (g.jac_to_affine(g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)) == expected).all()
    rb = g.register_bases(points=pts)
    try:
        assert (g.jac_to_affine(rb.multiexp_device(d_sc.data_ptr(), n, stream)) == expected).all()
        jac, err = rb.MultiExp(sc)
        assert err is None and (g.jac_to_affine(jac) == expected).all()
        m = 2 * 4096 + 1  # a prefix that ends one point into the third range
        jac, err = rb.MultiExp(sc[:m])
        assert err is None and (g.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all()
        with pytest.raises(RuntimeError, match="at most 2\\^27 points per ticket"):
            rb.submit(d_sc.data_ptr(), n)
    finally:
        rb.release()
    jac, err = g.MultiExp(pts, sc)
    assert err is None and (g.jac_to_affine(jac) == expected).all()
    jac, err = g.MultiExpSharded(pts, sc, devices=[0, 0], mode="windows")  # every rank splits its piece too
    assert err is None and (g.jac_to_affine(jac) == expected).all()


@pytest.mark.parametrize("ranges", [1, 3, 7])
def test_host_ranges_forced(gm, oracle_mod, forced_options, ranges):
    """The host entries cut a call into point ranges so that PCIe runs under compute; GMSM_OPT_HOST_RANGES forces the count."""
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = 10007
    pts, sc = _inputs(o, g, n, 6)
    expected = o.msm_affine(pts, sc, nthreads=8)
    forced_options(host_ranges=ranges)
    jac, err = g.MultiExp(pts, sc)
    assert err is None and (g.jac_to_affine(jac) == expected).all()
    rb = g.register_bases(points=pts)
    try:
        for m in (n, 5000, ranges):
            jac, err = rb.MultiExp(sc[:m])
            assert err is None and (g.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all(), m
    finally:
        rb.release()


def test_forced_window_width_is_clamped(gm, forced_options):
    """A forced width outside the documented 2..20 is refused (it used to be accepted up to 24 and then failed in hipMalloc)."""
    g = gm.G1Jac("bn254")
    with pytest.raises(ValueError, match="2..20"):
        gm.set_option("window_bits", 23)
    assert g.default_window_bits(1 << 20) == 16
    forced_options(window_bits=12)
    assert g.default_window_bits(1 << 20) == 12


def test_spreading_is_opt_in_and_gmsm_devices_is_parsed_strictly():
    """A process that configures nothing keeps every drop-in call on one device (no context anywhere else); GMSM_DEVICES
    opts in ("0,0" = two logical ranks on this box, "all"); a malformed or out-of-range list is an error of the call, not
    a silently shorter list.  Each case is its own process: the variable is read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import importlib, sys, numpy as np\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'oracle')!r})\n"
        "gm = importlib.import_module('gnark-crypto_amd'); import oracle\n"
        "g = gm.G1Jac('bn254'); o = oracle.Oracle('bn254', 'g1')\n"
        "n = (1 << 17) + 5\n"
        "pts = o.gen_points(n, 7, 3, nthreads=8)\n"
        "sc = np.random.default_rng(3).integers(0, 2**62, size=(n, 4), dtype=np.uint64)\n"
        "print('devices', gm.get_devices() if gm._lib.load().gmsm_get_devices(None, 0) >= 0 else 'error')\n"
        "jac, err = g.MultiExp(pts, sc)\n"
        "print('err', err)\n"
        "print('ok', err is None and bool((g.jac_to_affine(jac) == o.msm_affine(pts, sc, nthreads=16)).all()))\n")
    def run(value):
        env = {k: v for k, v in os.environ.items() if k != "GMSM_DEVICES"}
        if value is not None:
            env["GMSM_DEVICES"] = value
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout
    out = run(None)
    assert "devices [0]" in out and "ok True" in out
    out = run("0,0")
    assert "devices [0, 0]" in out and "ok True" in out
    out = run("all")
    assert "ok True" in out
    for bad in ("0,,1", "0,x", "0,", "7777"):
        out = run(bad)
        assert "devices error" in out and "GMSM_DEVICES" in out and "ok False" in out, (bad, out)
