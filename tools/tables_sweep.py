"""Window tables (gmsm_bases_precompute) against the plain registered-bases path: ms per MultiExp and stage times, the
table width swept.  usage: python tools/tables_sweep.py curve group logn [c ...]   (c = 0: the library's width)"""
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


def timed(lib, fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    lib.gmsm_set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = (ctypes.c_double * len(STAGES))()
    calls = ctypes.c_ulong(0)
    lib.gmsm_get_stage_times(st, len(STAGES), ctypes.byref(calls))
    lib.gmsm_set_profiling(0)
    return ms, " ".join(f"{s}={st[i] / max(1, calls.value):.3f}" for i, s in enumerate(STAGES[:-1])), out


def main():
    curve, group, logn = sys.argv[1], sys.argv[2], int(sys.argv[3])
    widths = [int(v) for v in sys.argv[4:]] or [0]
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    lib = gm._lib.load()
    n = 1 << logn
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2**64, size=(n, g.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (g.curve.fr_bits - 64 * (g.fr_limbs - 1) - 1)) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    sc = np.roll(a, 1, axis=0)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    reps = 8 if logn <= 22 else 3
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
    ms, st, ref = timed(lib, lambda: rb.multiexp_device(d_sc.data_ptr(), n, stream), reps)
    ms_h = 0.0 if os.environ.get("TS_DEVICE_ONLY") else timed(lib, lambda: rb.MultiExp(sc)[0], reps)[0]
    print(f"{curve} {group} 2^{logn} plain (c={g.default_window_bits(n)}): device scalars {ms:.3f} ms, host scalars {ms_h:.3f} ms | {st}", flush=True)
    ref = g.jac_to_affine(ref)
    rb.release()
    for c in widths:
        rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
        t0 = time.perf_counter()
        try:
            got = rb.precompute(c)
        except RuntimeError as e:
            print(f"  c={c}: {e}", flush=True)
            rb.release()
            continue
        pre_ms = (time.perf_counter() - t0) * 1e3
        try:
            ms, st, out = timed(lib, lambda: rb.multiexp_device(d_sc.data_ptr(), n, stream), reps)
            ms_h = 0.0 if os.environ.get("TS_DEVICE_ONLY") else timed(lib, lambda: rb.MultiExp(sc)[0], reps)[0]
            same = bool((g.jac_to_affine(out) == ref).all())
            print(f"  tables c={got} ({g.num_windows(got)} slabs, built in {pre_ms:.0f} ms): device scalars {ms:.3f} ms, host scalars {ms_h:.3f} ms, same={same} | {st}", flush=True)
        except RuntimeError as e:
            print(f"  c={got}: {e}", flush=True)
        rb.release()


if __name__ == "__main__":
    main()
