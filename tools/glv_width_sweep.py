#!/usr/bin/env python3
"""Per group and size: the best forced window width of the sorted pipeline without and with GLV half scalars (GMSM_OPT_GLV 0 / 2),
and what the library picks by itself (auto, GLV 1).   python tools/glv_width_sweep.py curve group logn[,logn...] [--steps=5]"""
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
    curve, group, logns = argv[0], argv[1], [int(x) for x in argv[2].split(",")]
    steps = 5
    for a in sys.argv[1:]:
        if a.startswith("--steps="):
            steps = int(a.split("=", 1)[1])
    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(0) == 0
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    nmax = 1 << max(logns)
    rng = np.random.default_rng([0x676C76, 7])
    a = bench.uniform_scalars(rng, g, nmax)
    b = bench.uniform_scalars(rng, g, nmax)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_b = torch.from_numpy(b.view(np.int64)).cuda()
    d_pts = torch.empty((nmax, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), nmax, d_pts.data_ptr(), stream)
    del d_a

    def ms(n):
        jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, jac

    print(f"== {curve} {group}: sorted pipeline, resident ms (mean of {steps}); per GLV setting every forced width, best first", flush=True)
    for logn in logns:
        n = 1 << logn
        ref = None
        row = [f"2^{logn}:"]
        lo = max(10, logn - 7)
        widths = [c for c in range(lo, min(18, logn + 2) + 1)]
        for glv in (0, 2):
            res = []
            for c in widths:
                with gm.options(glv=glv, window_bits=c, small_bits=1):
                    try:
                        t, jac = ms(n)
                    except RuntimeError:
                        continue
                aff = g.jac_to_affine(jac)
                if ref is None:
                    ref = aff
                assert (aff == ref).all(), (logn, glv, c)
                res.append((t, c))
            res.sort()
            row.append(f"glv{glv} " + " ".join(f"c{c}:{t:.3f}" for t, c in res[:4]) + " |")
        with gm.options(small_bits=1):
            t, jac = ms(n)
        assert (g.jac_to_affine(jac) == ref).all()
        row.append(f"auto {t:.3f}")
        print(" ".join(row), flush=True)


if __name__ == "__main__":
    main()
