"""Full-call sweep of the reduction geometry under the round-6 window counts (GLV halves the windows of most mid-size calls, so the
serial kernel of the reduction has half the threads the round-3/4 sweeps saw): GMSM_LOG2L forced 2..7 and GMSM_REDUCE_LEVELS 2 / 3
against the cost model's choice, resident ms per MultiExp and the reduce stage. Needs an -DGMSM_EXPERIMENTS build (tools/build_ab.sh,
GMSM_LIB=...); the shipped library ignores the switches.  -> profiles/r06_reduce_l_sweep.log
usage: GMSM_LIB=... python tools/reduce_l_sweep.py [curve group logn]..."""
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
    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(0) == 0
    args = sys.argv[1:]
    cfgs = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [("bw6_761", "g1", "20"), ("bn254", "g2", "20"), ("bls12_381", "g2", "22"),
                                                                      ("bls12_381", "g1", "22"), ("bn254", "g1", "20")]
    stream = torch.cuda.current_stream().cuda_stream
    for curve, group, logn in cfgs:
        logn = int(logn)
        g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
        n = 1 << logn
        rng = np.random.default_rng([0x72656475, logn])
        d_a = torch.from_numpy(bench.uniform_scalars(rng, g, n).view(np.int64)).cuda()
        d_b = torch.from_numpy(bench.uniform_scalars(rng, g, n).view(np.int64)).cuda()
        d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
        del d_a
        plan = g.default_plan(n)
        print(f"== {curve} {group} 2^{logn}: c = {plan['window_bits']}, {plan['windows']} windows, {plan['entries_per_point']} entries per point", flush=True)
        ref = None
        settings = [{}] + [{"GMSM_LOG2L": str(l)} for l in range(2, 8)] + [{"GMSM_REDUCE_LEVELS": "3"}, {"GMSM_REDUCE_LEVELS": "2"}, {}]
        for env in settings:
            for k in ("GMSM_LOG2L", "GMSM_REDUCE_LEVELS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                ms, jac = bench.loop_ms(lambda: g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream), 6, torch.cuda.synchronize, warm=2)
            except RuntimeError as e:
                print(f"  {env}: {e}", flush=True)
                continue
            prof = bench.StageProfile(lib)
            prof.start()
            for _ in range(4):
                g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
            torch.cuda.synchronize()
            st, _ = prof.stop()
            aff = g.jac_to_affine(jac)
            ref = aff if ref is None else ref
            assert (aff == ref).all(), env
            label = " ".join(f"{k[5:].lower()}={v}" for k, v in env.items()) or "auto"
            print(f"  {label:18s} {ms:8.4f} ms | reduce {st['reduce']:.3f} accumulate {st['accumulate']:.3f} fixup {st['fixup']:.3f}", flush=True)
        for k in ("GMSM_LOG2L", "GMSM_REDUCE_LEVELS"):
            os.environ.pop(k, None)
        del d_pts, d_b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
