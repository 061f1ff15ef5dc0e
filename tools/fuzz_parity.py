"""Randomised parity sweep (not part of the pytest suites): many sizes, scalar distributions and entry points against the
oracle; the registered-bases entries draw from three handles - plain, window tables of the library's width, tables of width 11.
usage: python tools/fuzz_parity.py [seconds] [curve] [g1|g2]"""
import importlib
import sys
import time

import numpy as np
import torch

import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "oracle"), os.path.join(_ROOT, "tests")):
    sys.path.insert(0, _p)
gm = importlib.import_module("gnark-crypto_amd")


def main():
    import oracle
    from conftest import random_scalars, scalars_from_ints
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    curve, which = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("bn254", "g1")
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle.Oracle(curve, which)
    rng = np.random.default_rng(20260924)
    nmax = 1 << 18
    pts_all = o.gen_points(nmax, 99, 5, nthreads=8)
    rb = g.register_bases(points=pts_all)
    # the same bases with window tables (gmsm_bases_precompute), the library's width and a narrow one; GMSM_OPT_TABLES = 2: every
    # call size runs through them (the default policy would send most of these sizes down the plain path)
    gm.set_option("tables", 2)
    handles = [rb, g.register_bases(points=pts_all), g.register_bases(points=pts_all)]
    handles[1].precompute(0)
    handles[2].precompute(11)
    t0 = time.time()
    cases = bad = trips = 0
    while time.time() - t0 < budget:
        n = int(rng.choice([rng.integers(1, 64), rng.integers(64, 5000), rng.integers(5000, nmax)]))
        kind = int(rng.integers(0, 11))
        # the fused small-n kernel on / off / forced widths and sizes; the experimental window-group split
        gm.set_option("small_bits", int(rng.choice([0, 0, 0, 1, 4, 5, 6, 7])))
        gm.set_option("small_max", int(rng.choice([0, 0, 300, 16384])))
        gm.set_option("split", int(rng.choice([0, 0, 0, 1])))
        # round 6: GLV half scalars never / by the measured table / always, the fused kernel's bucket phase by size / one lane / quads
        gm.set_option("glv", int(rng.choice([0, 1, 1, 2])))
        gm.set_option("small_quad", int(rng.choice([0, 0, 1, 2])))
        if kind == 0:
            sc = random_scalars(rng, g.curve, n)
        elif kind == 1:  # small values
            sc = scalars_from_ints(g.curve, [int(x) for x in rng.integers(0, 1 << 20, size=n)])
        elif kind == 2:  # few distinct values
            vals = [int.from_bytes(rng.bytes(32), "little") % g.curve.r for _ in range(3)]
            sc = scalars_from_ints(g.curve, [vals[int(i)] for i in rng.integers(0, 3, size=n)])
        elif kind == 3:  # all equal
            v = int.from_bytes(rng.bytes(32), "little") % g.curve.r
            sc = scalars_from_ints(g.curve, [v] * n)
        elif kind == 4:  # sparse: mostly zero
            sc = random_scalars(rng, g.curve, n)
            sc[rng.random(n) < 0.9] = 0
        elif kind == 5:  # powers of two and r - small
            sc = scalars_from_ints(g.curve, [(1 << int(e)) % g.curve.r if e >= 0 else g.curve.r + int(e)
                                             for e in rng.integers(-5, g.curve.fr_bits, size=n)])
        elif kind == 8:  # the reference's "smallvalues": every 5th STORED scalar = limbs [1, 0, ..] (multiexp_test.go:319-325)
            sc = random_scalars(rng, g.curve, n)
            sc[::5] = 0
            sc[::5, 0] = 1
        elif kind == 9:  # "redundancy": runs of 100 equal scalars (multiexp_test.go:327-334)
            sc = np.ascontiguousarray(np.repeat(random_scalars(rng, g.curve, (n + 99) // 100), 100, axis=0)[:n])
        elif kind == 10:  # 30 % of the scalars are the field element 1
            sc = random_scalars(rng, g.curve, n)
            sc[rng.random(n) < 0.3] = scalars_from_ints(g.curve, [1])[0]
        else:
            sc = random_scalars(rng, g.curve, n)
        pts = pts_all[:n]
        fresh_bases = False
        if kind == 6:  # one base repeated: P + P in buckets, partial sums and the reduction's combine
            pts = np.tile(pts_all[int(rng.integers(0, nmax)):][:1], (n, 1))
            fresh_bases = True
        elif kind == 7 and n >= 2:  # every base twice, with s and r - s: P - P everywhere, the total is infinity
            h = n // 2
            vals = [int.from_bytes(rng.bytes(32), "little") % g.curve.r for _ in range(h)]
            sc = scalars_from_ints(g.curve, vals + [(g.curve.r - v) % g.curve.r for v in vals] + [0] * (n - 2 * h))
            pts = np.concatenate([pts_all[:h], pts_all[:h], pts_all[:n - 2 * h]])
            fresh_bases = True
        want = o.msm_affine(pts, sc, nthreads=8)
        entry = 0 if fresh_bases and rng.random() < 0.5 else (4 if fresh_bases else int(rng.integers(0, 5)))
        if entry == 0:
            got, err = g.MultiExp(pts, sc)
            assert err is None
        elif entry == 4:  # the in-library multi-device entry, 2-5 logical ranks on device 0, either decomposition
            gj = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
            jac, err = gj.MultiExpSharded(pts, sc, devices=[0] * int(rng.integers(2, 6)), mode=str(rng.choice(["points", "windows"])))
            assert err is None, err
            got = g.jac_to_affine(jac)
        elif entry == 1:
            rb = handles[int(rng.integers(0, 3))]
            jac, err = rb.MultiExp(sc)
            assert err is None
            got = g.jac_to_affine(jac)
        elif entry == 2:
            d = torch.from_numpy(sc.view(np.int64)).cuda()
            torch.cuda.synchronize()
            rb = handles[int(rng.integers(0, 3))]
            got = g.jac_to_affine(rb.collect(rb.submit(d.data_ptr(), n)))
        else:
            rb = handles[int(rng.integers(0, 3))]
            jacs, err = rb.MultiExpBatch(scalars=np.stack([sc, sc]))
            assert err is None
            got = g.jac_to_affine(jacs[1])
        cases += 1
        if not (got == want).all():
            bad += 1
            print("MISMATCH", dict(n=n, kind=kind, entry=entry, glv=gm.get_option("glv"), quad=gm.get_option("small_quad")), flush=True)
        if cases % 16 == 0:  # round 6: the compressed encoding there and back on this case's bases (with the MSM result among them)
            both = np.concatenate([pts, got[None, :]])
            comp, err = g.Compress(points=both)
            assert err is None
            back, err = g.DecodeCompressed(comp)
            trips += 1
            if err is not None or not (back == both).all():
                bad += 1
                print("COMPRESSED ROUND TRIP MISMATCH", dict(n=n, err=err), flush=True)
    for k in ("small_bits", "small_max", "split", "small_quad"):
        gm.set_option(k, 0)
    gm.set_option("glv", 1)
    for h in handles:
        h.release()
    print(f"{curve} {which}: {cases} cases ({trips} compressed round trips), {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
