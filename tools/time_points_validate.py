"""Times gmsm_points_validate over resident points: level 2 (the reference's endomorphism identity) against level 3 ([r]P =
infinity, what rounds 1-5 ran for level 2).  python tools/time_points_validate.py [logn]"""
import ctypes
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
gm = importlib.import_module("gnark-crypto_amd")
lib = gm._lib.load()
logn_default = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for curve, which, logn in (("bls12_381", "g1", 20), ("bls12_381", "g2", 18), ("bw6_761", "g1", 18), ("bw6_761", "g2", 18), ("bn254", "g2", 20)):
    logn = logn_default or logn
    g = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2**62, size=(n, g.fr_limbs), dtype=np.uint64)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_p = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_p.data_ptr(), torch.cuda.current_stream().cuda_stream)
    bad = ctypes.c_int64(-1)
    ms = {}
    for level in (2, 3):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.gmsm_points_validate(g.gid, None, d_p.data_ptr(), n, level, ctypes.byref(bad))
            ts.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0, gm._lib.last_error()
        ms[level] = sorted(ts)[1]
    print(f"{curve} {which} 2^{logn}: identity {ms[2]:.2f} ms ({n / ms[2] / 1e3:.2f} M points/s), definition {ms[3]:.2f} ms, "
          f"ratio {ms[3] / ms[2]:.2f}", flush=True)
