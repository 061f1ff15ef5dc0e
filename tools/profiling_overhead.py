"""What the stage events cost: ms per resident MultiExp with gmsm_set_profiling(0) and (1), three rounds each.
usage: python tools/profiling_overhead.py"""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
g = gm.G1Jac("bn254"); lib = gm._lib.load()
stream = torch.cuda.current_stream().cuda_stream
for logn in (16, 20):
    n = 1 << logn
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, -1] &= np.uint64((1 << 60) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    torch.cuda.synchronize()
    def loop(k):
        t0 = time.perf_counter()
        for _ in range(k): g.multiexp_device(d_pts.data_ptr(), d_a.data_ptr(), n, stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    loop(5)
    res = []
    for rep in range(3):
        lib.gmsm_set_profiling(0); off = loop(40)
        lib.gmsm_set_profiling(1); on = loop(40)
        res.append((round(off, 4), round(on, 4)))
    lib.gmsm_set_profiling(0)
    print(f"2^{logn}: ms per call (profiling off, on):", res, flush=True)
