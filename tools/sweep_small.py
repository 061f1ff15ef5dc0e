#!/usr/bin/env python3
"""Fused small-n kernel: resident ms per call for every forced width against the sorted pipeline.
    python tools/sweep_small.py [curve group] [--logns=5,8,10,11,12,13]"""
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
    logns = (5, 8, 10, 11, 12, 13)
    for a in sys.argv[1:]:
        if a.startswith("--logns="):
            logns = tuple(int(x) for x in a.split("=", 1)[1].split(","))
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
            g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    cfg = gm.MultiExpConfig()
    for logn in logns:
        n = 1 << logn
        row = [f"2^{logn}:"]
        with gm.options(small_bits=1):
            row.append(f"pipeline {ms(n):.4f}")
            p_, s_ = np.ascontiguousarray(pts[:n]), np.ascontiguousarray(sc[:n])
            row.append(f"(cold {bench.median_ms(lambda: g.MultiExp(p_, s_, cfg), reps=9):.4f})")
        with gm.options(small_max=1 << 14):
            for c in range(3, 8):
                with gm.options(small_bits=c):
                    row.append(f"c{c} {ms(n):.4f}")
            with gm.options(small_bits=0):
                row.append(f"| auto {ms(n):.4f} (cold {bench.median_ms(lambda: g.MultiExp(p_, s_, cfg), reps=9):.4f})")
        print(" ".join(row), flush=True)
    t0 = time.perf_counter()
    for _ in range(200):
        g.fold_windows(np.zeros((g.num_windows(6), g.xyzz_limbs), dtype=np.uint64), 6)
    print(f"host fold of {g.num_windows(6)} windows at infinity: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms (call overhead)")


if __name__ == "__main__":
    main()
