"""Per-rank cost of the two multi-GPU decompositions, measured on ONE GPU (the rank-0 piece of each):
window sharding (all n points, windows 0, w, 2w, ...) against point sharding (n/w points, every window).
Used to pick the decomposition bench.py/sharding.py default to; prints one line per (logn, world)."""
import importlib
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")


def main():
    g = gm.G1Jac("bn254")
    for logn in (20, 24):
        n = 1 << logn
        pts = g.generate_points(n, 12345, 678)
        rng = np.random.default_rng(1)
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        sc[:, 3] %= np.uint64(0x3000000000000000)  # < r
        d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for world in (1, 2, 4, 8):
            def timed(fn, reps=5):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3
            c = g.default_window_bits(n)
            tw = timed(lambda: g.window_sums_device(d_pts.data_ptr(), d_sc.data_ptr(), n, c, 0, world))
            m = n // world
            cp = g.default_window_bits(m)
            tp = timed(lambda: g.window_sums_device(d_pts.data_ptr(), d_sc.data_ptr(), m, cp, 0, 1))
            print(f"logn={logn} world={world} window-shard piece {tw:.3f} ms (c={c})   point-shard piece {tp:.3f} ms (c={cp})", flush=True)


if __name__ == "__main__":
    main()
