"""Environment-switch sweep in ONE process: ms per MultiExp and per-stage times for a list of GMSM_* settings.
usage: python tools/sweep_env.py curve group logn [reps] -- "GMSM_C=20" "GMSM_C=17,GMSM_LOG2L=5,GMSM_PART_LOG2=14" ...
An empty string "" is the default configuration. GMSM_C / GMSM_TABLES / GMSM_MAX_RUN / GMSM_HOST_RANGES are applied with
gmsm_set_option (the library reads the environment once, at load); the tuning knobs exist only in a -DGMSM_EXPERIMENTS
build (tools/build_ab.sh + GMSM_LIB), which reads them from the environment per call. One set of device-resident inputs
serves every variant. Every variant's affine result is compared with the default's."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
OPTION_OF = {"GMSM_C": "window_bits", "GMSM_TABLES": "tables", "GMSM_MAX_RUN": "max_run", "GMSM_HOST_RANGES": "host_ranges"}
STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "reserved"]


def main():
    sep = sys.argv.index("--")
    curve, group, logn = sys.argv[1], sys.argv[2], int(sys.argv[3])
    reps = int(sys.argv[4]) if sep > 4 else 4
    variants = sys.argv[sep + 1:]
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
    d_sc = torch.from_numpy(np.roll(a, 1, axis=0).view(np.int64)).cuda()
    torch.cuda.synchronize()
    ref = None
    for var in variants:
        keys = []
        for kv in filter(None, var.split(",")):
            k, v = kv.split("=")
            if k in OPTION_OF:
                gm.set_option(OPTION_OF[k], int(v))
            else:
                os.environ[k] = v
            keys.append(k)
        try:
            out = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            torch.cuda.synchronize()
            aff = g.jac_to_affine(out)
            if ref is None:
                ref = aff
            same = bool((aff == ref).all())
            lib.gmsm_set_profiling(1)
            t0 = time.perf_counter()
            for _ in range(reps):
                g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            st = (ctypes.c_double * len(STAGES))()
            calls = ctypes.c_ulong(0)
            lib.gmsm_get_stage_times(st, len(STAGES), ctypes.byref(calls))
            lib.gmsm_set_profiling(0)
            nc = max(1, calls.value)
            stages = " ".join(f"{s[:4]}={st[i] / nc:.3f}" for i, s in enumerate(STAGES[:-1]))
            print(f"{curve} {group} 2^{logn} [{var or 'default'}] {ms:.3f} ms (profiled) same={same} | {stages}", flush=True)
        except Exception as e:  # a variant the library refuses (geometry limits) must not end the sweep
            print(f"{curve} {group} 2^{logn} [{var}] FAILED: {e}", flush=True)
        for k in keys:
            if k in OPTION_OF:
                gm.set_option(OPTION_OF[k], 1 if k == "GMSM_TABLES" else 0)
            else:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
