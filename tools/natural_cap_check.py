"""One MultiExp ABOVE the natural pipeline-run cap (2^27 points) with no environment override: 33 blocks of 2^22
(point, scalar) pairs = 2^27 + 2^22 points, so gmsm_multiexp_device cuts the call into two point ranges by itself
(Group::device_ranges, the reference's split + AddAssign, ecc/bn254/multiexp.go:98-140).  Checked through the closed form
of tests/test_gpu_parity.py::test_bn254_g1_2_pow_26_closed_form: bases [a_i]G, result must be [33 * sum a_i b_i]G.
Needs ~25 GB of device memory; not part of the suite (the suite covers the same path with GMSM_OPT_MAX_RUN lowered).
Usage: python tools/natural_cap_check.py        (GPU box; the oracle is the checker)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import importlib
    import torch
    import oracle as oracle_mod
    from conftest import random_scalars, rng_for
    gm = importlib.import_module("gnark-crypto_amd")
    assert gm.get_option("max_run") == 0
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    m, reps = 1 << 22, 33
    rng = rng_for(27, 1)
    a = random_scalars(rng, g.curve, m)
    b = random_scalars(rng, g.curve, m)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_blk = torch.empty((m, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), m, d_blk.data_ptr(), stream)
    d_pts = d_blk.repeat(reps, 1).contiguous()
    d_sc = torch.from_numpy(b.view(np.int64)).cuda().repeat(reps, 1).contiguous()
    n = m * reps
    assert n > 1 << 27
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        jac = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
        ms.append((time.perf_counter() - t0) * 1e3)
    fr = oracle_mod.Field("bn254_fr", 4)
    k_limbs = fr.from_mont(fr.dot(a, b))
    k = sum(int(v) << (64 * i) for i, v in enumerate(k_limbs)) * reps % g.curve.r
    expected = o.jac_to_affine(o.scalar_mul(o.generator, k))
    ok = bool((g.jac_to_affine(jac) == expected).all())
    print(f"n = 2^27 + 2^22 = {n} points, point ranges cut by the library itself: parity {'OK' if ok else 'MISMATCH'}; "
          f"ms per call {', '.join(f'{v:.1f}' for v in ms)} (first includes workspace growth)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
