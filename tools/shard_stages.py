"""Stage times of ONE rank's piece of a sharded MultiExp, measured on one GPU (BN254 G1): the window decomposition (all n
points, windows r, r + world, ...) and the point decomposition (n / world points, every window) for world = 1, 2, 4, 8 -
where a rank's time goes once the accumulation has been divided by the rank count (profiles/rNN_shard_stages.log)."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "reserved"]


def staged(lib, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    lib.gmsm_set_profiling(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    st = (ctypes.c_double * len(STAGES))()
    calls = ctypes.c_ulong(0)
    lib.gmsm_get_stage_times(st, len(STAGES), ctypes.byref(calls))
    lib.gmsm_set_profiling(0)
    nc = max(1, calls.value)
    return ms, " ".join(f"{s[:5]}={st[i] / nc:.3f}" for i, s in enumerate(STAGES[:-1]))


def main():
    g = gm.G1Jac("bn254")
    lib = gm._lib.load()
    for logn in (20, 24):
        n = 1 << logn
        rng = np.random.default_rng(1)
        a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        d_a = torch.from_numpy(a.view(np.int64)).cuda()
        d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
        d_sc = torch.from_numpy(np.roll(a, 1, axis=0).view(np.int64)).cuda()
        for world in (1, 2, 4, 8):
            c = g.default_window_bits(n)
            ms, st = staged(lib, lambda: g.window_sums_device(d_pts.data_ptr(), d_sc.data_ptr(), n, c, 0, world, stream))
            print(f"2^{logn} world={world} windows piece (c={c}): {ms:.3f} ms | {st}", flush=True)
            m = n // world
            cp = g.default_window_bits(m)
            ms, st = staged(lib, lambda: g.window_sums_device(d_pts.data_ptr(), d_sc.data_ptr(), m, cp, 0, 1, stream))
            print(f"2^{logn} world={world} points  piece (c={cp}): {ms:.3f} ms | {st}", flush=True)
        del d_pts, d_sc, d_a
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
