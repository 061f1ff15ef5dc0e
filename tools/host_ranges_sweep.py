"""Cold / warm-bases time of the host-pointer entries for a list of GMSM_OPT_HOST_RANGES settings (0 = the library's choice).
usage: python tools/host_ranges_sweep.py curve group logn ranges,ranges,...   e.g.  bn254 g1 20 0,1,2,4,8"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")


def median_ms(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    curve, group, logn = sys.argv[1], sys.argv[2], int(sys.argv[3])
    variants = [int(v) for v in sys.argv[4].split(",")]
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng(11)
    a = rng.integers(0, 2**64, size=(n, g.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (g.curve.fr_bits - 64 * (g.fr_limbs - 1) - 1)) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    pts = d_pts.cpu().numpy().view(np.uint64)
    sc = np.roll(a, 1, axis=0).copy()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    ref = g.jac_to_affine(g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream))
    resident = median_ms(lambda: g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream), 5)
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
    reps = 5 if logn <= 22 else 3
    print(f"{curve} {group} 2^{logn}: resident {resident:.3f} ms", flush=True)
    for v in variants:
        gm.set_option("host_ranges", v or 0)
        jc, err = g.MultiExp(pts, sc)
        assert err is None, err
        jw, err = rb.MultiExp(sc)
        assert err is None, err
        ok = bool((g.jac_to_affine(jc) == ref).all() and (g.jac_to_affine(jw) == ref).all())
        cold = median_ms(lambda: g.MultiExp(pts, sc), reps)
        warm = median_ms(lambda: rb.MultiExp(sc), reps)
        print(f"  ranges={v or 'auto':>4}: cold {cold:8.3f} ms   warm-bases {warm:8.3f} ms   same={ok}", flush=True)
    rb.release()


if __name__ == "__main__":
    main()
