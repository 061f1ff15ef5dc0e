"""K MultiExp calls over registered bases with window tables and nothing else on the device after the set-up: the command
rocprofv3 profiles for the tables block of profiles/r03_kernel_stats.md.
usage: python tools/tables_profile.py curve group logn [calls]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")


def main():
    curve, group, logn = sys.argv[1], sys.argv[2], int(sys.argv[3])
    calls = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2**64, size=(n, g.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (g.curve.fr_bits - 64 * (g.fr_limbs - 1) - 1)) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    d_sc = torch.from_numpy(np.roll(a, 1, axis=0).view(np.int64)).cuda()
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
    c = rb.precompute(0)
    for _ in range(calls):
        rb.multiexp_device(d_sc.data_ptr(), n, stream)
    torch.cuda.synchronize()
    print(f"{curve} {group} 2^{logn}: {calls} calls through window tables of width {c}")
    rb.release()


if __name__ == "__main__":
    main()
