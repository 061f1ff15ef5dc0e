"""Throughput of the fixed-base batch (gmsm_batch_scalar_mul_device): n random scalars times the generator, scalars and
results resident in HBM. Prints one JSON line per size; the CPU figure is the oracle's double-and-add on a sample."""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
sys.path.insert(0, "tests")
gm = importlib.import_module("gnark-crypto_amd")


def main():
    import oracle
    from conftest import random_scalars, rng_for
    curve, which = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("bn254", "g1")
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle.Oracle(curve, which)
    for logn in (16, 20, 22):
        n = 1 << logn
        sc = random_scalars(rng_for(5, logn), g.curve, n)
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        d_out = torch.zeros((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        g.batch_scalar_mul_device(o.generator, d_sc.data_ptr(), n, d_out.data_ptr())
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            g.batch_scalar_mul_device(o.generator, d_sc.data_ptr(), n, d_out.data_ptr())
        dt = (time.perf_counter() - t0) / reps
        # spot-check against the oracle and time the oracle's double-and-add on a sample
        res = d_out[:64].cpu().numpy().view(np.uint64)
        Fr = oracle.Field(f"{g.curve.name}_fr", g.curve.fr_limbs)
        t1 = time.perf_counter()
        ok = True
        for i in range(64):
            k = sum(int(v) << (64 * j) for j, v in enumerate(Fr.from_mont(sc[i])))
            ok = ok and bool((o.jac_to_affine(o.scalar_mul(o.generator, k)) == res[i]).all())
        cpu_per = (time.perf_counter() - t1) / 64
        print(json.dumps({"op": "BatchScalarMultiplication", "group": f"{curve} {which}", "n": n, "ms": round(dt * 1e3, 3),
                          "scalar_muls_per_s": round(n / dt), "bit_exact_sample": ok,
                          "cpu_oracle_double_and_add_us_each_1thread": round(cpu_per * 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
