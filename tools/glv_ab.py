#!/usr/bin/env python3
"""Same-process A/B of GLV half scalars in the sorted pipeline (GMSM_OPT_GLV 0 / 2): resident ms per MultiExp and the stage
times, per window width.   python tools/glv_ab.py [curve group logn] [--widths=15,16,17,18] [--steps=10]"""
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
    curve, group, logn = (argv + ["bn254", "g1", "20"])[:3]
    logn = int(logn)
    widths, steps = (0,), 10
    for a in sys.argv[1:]:
        if a.startswith("--widths="):
            widths = tuple(int(x) for x in a.split("=", 1)[1].split(","))
        if a.startswith("--steps="):
            steps = int(a.split("=", 1)[1])
    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(0) == 0
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng([0x676C76, logn])
    a = bench.uniform_scalars(rng, g, n)
    b = bench.uniform_scalars(rng, g, n)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_b = torch.from_numpy(b.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    del d_a
    ref = None
    print(f"== {curve} {group} 2^{logn}: resident ms per MultiExp (mean of {steps}), stage ms", flush=True)
    for c in widths:
        for glv in (0, 2, 0, 2):
            with gm.options(glv=glv, window_bits=c):
                jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                prof = bench.StageProfile(lib)
                prof.start()
                for _ in range(max(2, steps // 2)):
                    g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
                torch.cuda.synchronize()
                st, _ = prof.stop()
            aff = g.jac_to_affine(jac)
            if ref is None:
                ref = aff
            assert (aff == ref).all(), (c, glv)
            print(f"c={c or 'auto'} glv={glv}: {ms:.4f} ms | " + " ".join(f"{k} {v:.3f}" for k, v in st.items() if k != "reserved"), flush=True)


if __name__ == "__main__":
    main()
