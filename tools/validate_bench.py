import importlib, time, ctypes, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
gm = importlib.import_module("gnark-crypto_amd")
lib = gm._lib.load()
for curve, which, logn in (("bls12_381", "g1", 20), ("bls12_381", "g2", 18), ("bw6_761", "g1", 18), ("bn254", "g2", 20)):
    g = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2**62, size=(n, g.fr_limbs), dtype=np.uint64)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_p = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_p.data_ptr(), torch.cuda.current_stream().cuda_stream)
    bad = ctypes.c_int64(-1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.gmsm_points_validate(g.gid, None, d_p.data_ptr(), n, 2, ctypes.byref(bad))
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, gm._lib.last_error()
    print(f"{curve} {which} 2^{logn} validate level 2: {sorted(ts)[1]:.2f} ms ({n / sorted(ts)[1] / 1e3:.2f} M points/s)", flush=True)
