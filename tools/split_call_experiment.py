#!/usr/bin/env python3
"""GMSM_OPT_SPLIT: the windows of ONE MultiExp in two groups - the fix-up + reduction of the first group on a second stream
beside the accumulation of the second group (round 2 measured a three-stream form of this as slower).
Alternates split off / on in one process, same resident inputs; prints ms per call, the stage times of the timed-outside
pass, and whether the affine results agree.   python tools/split_call_experiment.py [logn ...]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "reserved"]


def main():
    import torch
    import bench
    logns = [int(a) for a in sys.argv[1:]] or [20, 22, 24]
    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(0) == 0
    g = gm.G1Jac("bn254")
    for logn in logns:
        n = 1 << logn
        rng = np.random.default_rng(11)
        a = bench.uniform_scalars(rng, g, n)
        sc = bench.uniform_scalars(rng, g, n)
        d_a = torch.from_numpy(a.view(np.int64)).cuda()
        d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        reps = 30 if logn <= 20 else 8
        ref = None
        for rnd in range(2):
            for split in (0, 1):
                with gm.options(split=split):
                    out = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        out = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / reps * 1e3
                    lib.gmsm_set_profiling(1)
                    for _ in range(4):
                        g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
                    torch.cuda.synchronize()
                    st = (ctypes.c_double * len(STAGES))()
                    calls = ctypes.c_ulong(0)
                    lib.gmsm_get_stage_times(st, len(STAGES), ctypes.byref(calls))
                    lib.gmsm_set_profiling(0)
                    nc = max(1, calls.value)
                aff = g.jac_to_affine(out)
                ref = aff if ref is None else ref
                print(f"2^{logn} split={split} run {rnd}: {ms:.4f} ms  same={bool((aff == ref).all())} | "
                      + " ".join(f"{s[:4]}={st[i] / nc:.3f}" for i, s in enumerate(STAGES[:-1])), flush=True)
        del d_pts, d_sc, d_a
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
