"""Window-size sweep: ms per MultiExp for every c in a range, sizes 2^lo..2^hi (BN254 G1 unless told otherwise).
usage: python tools/sweep_c.py [lo hi [curve group [cmin cmax]]]   (the width is forced with gmsm_set_option(GMSM_OPT_WINDOW_BITS))"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
gm = importlib.import_module("gnark-crypto_amd")


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 21)
    curve, group = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("bn254", "g1")
    cmin, cmax = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (7, 16)
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    nmax = 1 << hi
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2**64, size=(nmax, g.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (g.curve.fr_bits - 64 * (g.fr_limbs - 1) - 1)) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((nmax, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), nmax, d_pts.data_ptr(), stream)
    d_sc = torch.from_numpy(np.roll(a, 1, axis=0).view(np.int64)).cuda()
    for logn in range(lo, hi + 1):
        n = 1 << logn
        gm.set_option("window_bits", 0)
        default_c = g.default_window_bits(n)
        row = {}
        for c in range(cmin, cmax + 1):
            gm.set_option("window_bits", c)
            try:
                g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            except RuntimeError as e:  # a width the pipeline refuses for this many points (32-bit sort entries): skip it
                print(f"  2^{logn} c={c}: skipped ({str(e)[:90]})", flush=True)
                continue
            torch.cuda.synchronize()
            reps = 8 if logn <= 18 else 4
            t0 = time.perf_counter()
            for _ in range(reps):
                g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            torch.cuda.synchronize()
            row[c] = (time.perf_counter() - t0) / reps * 1e3
        if not row:
            continue
        best = min(row, key=row.get)
        print(f"{curve} {group} 2^{logn}: default c={default_c} {row.get(default_c, float('nan')):.3f} ms | best c={best} {row[best]:.3f} ms | " +
              " ".join(f"{c}:{v:.3f}" for c, v in row.items()), flush=True)
    gm.set_option("window_bits", 0)


if __name__ == "__main__":
    main()
