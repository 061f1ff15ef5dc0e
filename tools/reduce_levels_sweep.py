"""Per-rank pieces of a window-sharded MultiExp under forced reduction geometry (GMSM_REDUCE_LEVELS / GMSM_LOG2L: an
-DGMSM_EXPERIMENTS build, tools/build_ab.sh, loaded with GMSM_LIB; the shipped library ignores the switches and prints its
own choice in every column) -> profiles/r03_reduce_levels.log.  usage: [GMSM_LIB=...] python tools/reduce_levels_sweep.py"""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "reserved"]
g = gm.G1Jac("bn254"); lib = gm._lib.load()
def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    lib.gmsm_set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = (ctypes.c_double * len(STAGES))(); calls = ctypes.c_ulong(0)
    lib.gmsm_get_stage_times(st, len(STAGES), ctypes.byref(calls)); lib.gmsm_set_profiling(0)
    return ms, st[6] / max(1, calls.value)
for logn in (20, 24):
    n = 1 << logn
    pts = g.generate_points(n, 12345, 678)
    rng = np.random.default_rng(1)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); sc[:, 3] %= np.uint64(0x3000000000000000)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda(); d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    for world in (2, 4, 8):
        c = g.default_window_bits(n)
        out = []
        for env in ({}, {"GMSM_REDUCE_LEVELS": "3"}, {"GMSM_REDUCE_LEVELS": "3", "GMSM_LOG2L": "1"}, {"GMSM_REDUCE_LEVELS": "3", "GMSM_LOG2L": "2"}, {"GMSM_REDUCE_LEVELS": "2", "GMSM_LOG2L": "2"}, {"GMSM_REDUCE_LEVELS": "2", "GMSM_LOG2L": "3"}):
            for k in ("GMSM_REDUCE_LEVELS", "GMSM_LOG2L"): os.environ.pop(k, None)
            os.environ.update(env)
            try:
                ms, red = timed(lambda: g.window_sums_device(d_pts.data_ptr(), d_sc.data_ptr(), n, c, 0, world))
                out.append(f"{'+'.join(v for v in env.values()) or 'auto'}: {ms:.3f} (reduce {red:.3f})")
            except RuntimeError as e:
                out.append(f"{env}: {e}")
        print(f"2^{logn} window-sharded piece of {world} (c={c}), levels+log2L: " + " | ".join(out), flush=True)
