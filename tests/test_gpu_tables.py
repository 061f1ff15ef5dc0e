"""Window tables of registered bases (gmsm_bases_precompute): the multiples 2^(c w) P_i live in HBM and every MultiExp
over the handle fills ONE bucket set - same group element as the plain path, so every result must equal the oracle's
affine limbs bit for bit.  Covers every entry that takes a handle (host and device scalars, tickets, batches, sharded
handles), prefixes (long ones through the tables, short ones through the plain path), point-range splits of the shared
bucket set, crowded buckets, and the three-level bucket reduction that one set of 2^(c-1) buckets runs.
Reference shape: MultiExp over a fixed SRS, kzg.Commit (ecc/bn254/kzg/kzg.go:159-176) over multiexp.go:61-140."""
import numpy as np
import pytest

from conftest import random_scalars, rng_for, scalars_from_ints

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def tables_for_every_size(gm):
    """The library sends only the call sizes through the tables for which they were measured to win (2^13..2^21 points,
    by group); the parity tests are smaller: GMSM_OPT_TABLES = 2 = whenever the handle has tables."""
    with gm.options(tables=2):
        yield


def table_runs(gm):
    return int(gm._lib.load().gmsm_debug_table_runs())


def _jac_group(gm, curve, which):
    return (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)


def _inputs(o, g, n, tag):
    rng = rng_for(41, g.gid, tag, n & 0xFFFFFF)
    pts = o.gen_points(n, int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62)), nthreads=16)
    sc = random_scalars(rng, g.curve, n)
    pts[[3, n // 2, n - 1]] = 0   # points at infinity
    sc[[7, n // 3]] = 0           # zero scalars
    sc[::13, 1:] = 0              # single-limb scalars: the upper windows see digit 0 only
    return pts, sc


@pytest.mark.parametrize("curve,which,n,widths", [("bn254", "g1", (1 << 17) + 77, (0, 11, 17)),
                                                   ("bls12_381", "g1", (1 << 16) + 5, (0, 16)),
                                                   ("bn254", "g2", (1 << 15) + 9, (0,)),
                                                   ("bls12_381", "g2", (1 << 14) + 3, (13,)),
                                                   ("bw6_761", "g1", (1 << 14) + 1, (0, 12))])
def test_tables_every_entry(gm, oracle_mod, curve, which, n, widths):
    import torch
    g = _jac_group(gm, curve, which)
    o = oracle_mod.Oracle(curve, which)
    pts, sc = _inputs(o, g, n, 1)
    expected = o.msm_affine(pts, sc, nthreads=16)
    stream = torch.cuda.current_stream().cuda_stream
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    for c in widths:
        rb = g.register_bases(points=pts)
        try:
            assert rb.table_bits == 0
            got_c = rb.precompute(c)
            assert got_c == (c or rb.table_bits) and 2 <= got_c <= 20
            assert rb.precompute(0) == got_c          # idempotent
            with pytest.raises(RuntimeError, match="another width"):
                rb.precompute(got_c - 1)
            # host scalars: all bases, prefixes through the tables (>= n/16), short prefixes through the plain path
            for m in (n, n - 1, n // 2 + 1, n // 16 + 1, 300, 1, 0):
                before = table_runs(gm)
                jac, err = rb.MultiExp(sc[:m])
                assert err is None, err
                assert (g.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=16)).all(), (c, m)
                assert (table_runs(gm) > before) == (m > 0), (c, m)
            # device scalars, ticket pair, batch
            before = table_runs(gm)
            assert (g.jac_to_affine(rb.multiexp_device(d_sc.data_ptr(), n, stream)) == expected).all()
            t1 = rb.submit(d_sc.data_ptr(), n, stream)
            t2 = rb.submit(d_sc.data_ptr(), n // 2, stream)
            assert (g.jac_to_affine(rb.collect(t1)) == expected).all()
            assert (g.jac_to_affine(rb.collect(t2)) == o.msm_affine(pts[: n // 2], sc[: n // 2], nthreads=16)).all()
            m = min(n, 6000)
            vecs = np.stack([random_scalars(rng_for(43, j), g.curve, m) for j in range(3)])
            jacs, err = rb.MultiExpBatch(scalars=vecs)
            assert err is None, err
            for j in range(3):
                assert (g.jac_to_affine(jacs[j]) == o.msm_affine(pts[:m], vecs[j], nthreads=8)).all(), j
            assert table_runs(gm) == before + 6
            _, err = rb.MultiExp(np.zeros((n + 1, g.fr_limbs), dtype=np.uint64))
            assert err == "len(points) != len(scalars)"
        finally:
            rb.release()


def test_tables_crowded_buckets(gm, oracle_mod):
    """Scalar distributions that put most entries of the shared set into few buckets (multiexp_test.go:319-334
    'smallvalues' / 'redundancy', every scalar equal, +-1): long chains of partial sums, the k_fixup_long path."""
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = (1 << 16) + 11
    pts, _ = _inputs(o, g, n, 2)
    rb = g.register_bases(points=pts)
    try:
        rb.precompute(0)
        rng = rng_for(44)
        r = g.curve.r
        cases = {
            "all equal": scalars_from_ints(g.curve, [0x1234567 * (1 << 100) + 5] * n),
            "small": scalars_from_ints(g.curve, [int(v) for v in rng.integers(0, 8, size=n)]),
            "plus minus one": scalars_from_ints(g.curve, [1 if i & 1 else r - 1 for i in range(n)]),
            "r-1": scalars_from_ints(g.curve, [r - 1] * n),
            "few distinct": scalars_from_ints(g.curve, [int(v) * 0x9E3779B97F4A7C15F39CC0605CEDC835 % r
                                                        for v in rng.integers(1, 5, size=n)]),
        }
        for name, sc in cases.items():
            jac, err = rb.MultiExp(sc)
            assert err is None, err
            assert (g.jac_to_affine(jac) == o.msm_affine(pts, sc, nthreads=16)).all(), name
    finally:
        rb.release()


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bls12_381", "g2")])
def test_tables_point_ranges_and_switches(gm, oracle_mod, curve, which):
    """The shared bucket set under the splits of the entries: pipeline-run cap (device ranges + k_merge_buckets), forced
    host ranges; GMSM_OPT_TABLES = 0 and a foreign forced width fall back to the plain path over the same handle."""
    import torch
    g = _jac_group(gm, curve, which)
    o = oracle_mod.Oracle(curve, which)
    n = 3 * 4096 + 77
    pts, sc = _inputs(o, g, n, 3)
    expected = o.msm_affine(pts, sc, nthreads=8)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    rb = g.register_bases(points=pts)
    try:
        c = rb.precompute(9)
        assert c == 9
        for env in ({"max_run": 4096}, {"host_ranges": 5}, {"tables": 0}, {"window_bits": 11},
                    {"window_bits": 9, "max_run": 5000}):
            with gm.options(**env):
                before = table_runs(gm)
                assert (g.jac_to_affine(rb.multiexp_device(d_sc.data_ptr(), n, stream)) == expected).all(), env
                plain = env.get("tables") == 0 or env.get("window_bits") == 11
                ranges = 4 if env.get("max_run") == 4096 else 3 if "max_run" in env else 1
                assert table_runs(gm) - before == (0 if plain else ranges), env
                jac, err = rb.MultiExp(sc)
                assert err is None and (g.jac_to_affine(jac) == expected).all(), env
                m = 2 * 4096 + 1
                jac, err = rb.MultiExp(sc[:m])
                assert err is None and (g.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=8)).all(), env
    finally:
        rb.release()


def test_tables_on_sharded_handle(gm, oracle_mod):
    """gmsm_bases_precompute on a handle of gmsm_bases_register_sharded: tables on every replica; point slices run
    through them (one bucket set per rank), window slices and short prefixes through the plain path."""
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = (1 << 18) + 77
    pts, sc = _inputs(o, g, n, 4)
    rb = g.register_bases_sharded(pts, devices=[0, 0, 0])
    try:
        c = rb.precompute(0)
        assert c == rb.table_bits and c >= 10
        for m in (n, (1 << 17) + 1, 3000):
            exp = o.msm_affine(pts[:m], sc[:m], nthreads=16)
            before = table_runs(gm)
            jac, err = rb.MultiExp(sc[:m])
            assert err is None and (g.jac_to_affine(jac) == exp).all(), m
            assert table_runs(gm) > before
            for mode in ("points", "windows"):
                jac, err = rb.MultiExpSharded(sc[:m], mode=mode)
                assert err is None and (g.jac_to_affine(jac) == exp).all(), (m, mode)
        m = 40000
        vecs = np.stack([random_scalars(rng_for(45, j), g.curve, m) for j in range(4)])
        jacs, err = rb.MultiExpBatch(scalars=vecs)
        assert err is None, err
        for j in range(4):
            assert (g.jac_to_affine(jacs[j]) == o.msm_affine(pts[:m], vecs[j], nthreads=8)).all(), j
    finally:
        rb.release()


def test_tables_default_call_sizes(gm, oracle_mod, forced_options):
    """Without the test switch the tables serve the measured ranges only: a 2^13-point call over BN254 G1 runs through the
    wide tables, calls of at most 2^12 points through the narrow ones (the fused small-n kernel, round 5), a call between
    the two ranges and a prefix below n/16 take the plain paths; results agree either way."""
    forced_options(tables=1)
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = (1 << 16) + 3
    pts, sc = _inputs(o, g, n, 5)
    rb = g.register_bases(points=pts)
    try:
        assert rb.precompute(0) == 16
        for m, through in ((n, True), (1 << 13, True), ((1 << 13) - 1, False), (n // 16 - 1, True), (1 << 12, True), ((1 << 12) + 1, False)):
            before = table_runs(gm)
            jac, err = rb.MultiExp(sc[:m])
            assert err is None and (g.jac_to_affine(jac) == o.msm_affine(pts[:m], sc[:m], nthreads=16)).all(), m
            assert (table_runs(gm) > before) == through, m
    finally:
        rb.release()


def test_tables_concurrent_callers(gm, oracle_mod):
    """Several threads over one handle with tables at once (two workspaces per device, the callers queue for them): blocking
    calls with host scalars, device scalars and tickets interleaved; every result equals the oracle's."""
    import threading
    import torch
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = (1 << 15) + 7
    pts, sc = _inputs(o, g, n, 6)
    expected = o.msm_affine(pts, sc, nthreads=16)
    half = o.msm_affine(pts[: n // 2], sc[: n // 2], nthreads=16)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    rb = g.register_bases(points=pts)
    try:
        rb.precompute(0)
        before = table_runs(gm)
        bad = []

        def host_caller(i):
            for k in range(4):
                m = n if (i + k) % 2 == 0 else n // 2
                jac, err = rb.MultiExp(sc[:m])
                if err is not None or not (g.jac_to_affine(jac) == (expected if m == n else half)).all():
                    bad.append(("host", i, k, err))

        def device_caller(i):
            for k in range(4):
                jac = rb.multiexp_device(d_sc.data_ptr(), n, 0)
                if not (g.jac_to_affine(jac) == expected).all():
                    bad.append(("device", i, k))

        th = [threading.Thread(target=host_caller, args=(i,)) for i in range(3)]
        th += [threading.Thread(target=device_caller, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad
        assert table_runs(gm) - before == 20
    finally:
        rb.release()
