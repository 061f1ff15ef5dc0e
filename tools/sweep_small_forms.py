#!/usr/bin/env python3
"""Fused small-n kernel, same process A/B of its forms: resident ms per call for GLV half scalars off / on x the bucket phase on
one lane per entry / on lane quads, per forced window width, against the library's own choice.
    python tools/sweep_small_forms.py [curve group] [--logns=5,8,10,11,12,13] [--widths=4,5,6,7]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    curve, group = (argv + ["bn254", "g1"])[:2]
    logns, widths = (5, 8, 10, 11, 12, 13), (5, 6, 7)
    for a in sys.argv[1:]:
        if a.startswith("--logns="):
            logns = tuple(int(x) for x in a.split("=", 1)[1].split(","))
        if a.startswith("--widths="):
            widths = tuple(int(x) for x in a.split("=", 1)[1].split(","))
    gm = importlib.import_module("gnark-crypto_amd")
    assert gm._lib.load().gmsm_set_device(0) == 0
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    nmax = 1 << max(logns)
    rng = np.random.default_rng(7)
    pts = g.generate_points(nmax, 12345, 678)
    sc = bench.uniform_scalars(rng, g, nmax)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def ms(n, reps=40):
        g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    print(f"== {curve} {group}: resident ms per call; forms: glv0/glv1 = GLV half scalars off/on, lane/quad = bucket phase", flush=True)
    for logn in logns:
        n = 1 << logn
        with gm.options(small_bits=1):
            ref_ms, ref = ms(n)
        row = [f"2^{logn}: pipeline {ref_ms:.4f} |"]
        ref_aff = g.jac_to_affine(ref)
        with gm.options(small_max=1 << 13):
            for glv in (0, 1):
                for quad in (1, 2):
                    best = None
                    for c in widths:
                        with gm.options(small_bits=c, glv=glv, small_quad=quad):
                            t, out = ms(n)
                        assert (g.jac_to_affine(out) == ref_aff).all(), (logn, glv, quad, c)
                        if best is None or t < best[0]:
                            best = (t, c)
                    row.append(f"glv{glv}/{'quad' if quad == 2 else 'lane'} {best[0]:.4f} (c{best[1]})")
            t, out = ms(n)
            assert (g.jac_to_affine(out) == ref_aff).all()
            row.append(f"| auto {t:.4f}")
        print(" ".join(row), flush=True)


if __name__ == "__main__":
    main()
