"""Host-side window fold (gmsm_fold_windows) alone, microseconds per call for four groups - needs no GPU.
usage: python tools/fold_time.py"""
import importlib, time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
for curve, which in (("bn254","g1"),("bls12_381","g1"),("bn254","g2"),("bw6_761","g1")):
    g = (gm.G1Jac if which=="g1" else gm.G2Jac)(curve)
    c = 16; nwin = g.num_windows(c)
    pts = g.generate_points(nwin, 3, 5)
    L = g.coord_limbs
    base = (g.curve.p.bit_length() + 63) // 64
    R = (1 << (64*base)) % g.curve.p
    one = np.zeros(L, dtype=np.uint64); one[:base] = [(R >> (64*i)) & (2**64-1) for i in range(base)]
    tot = np.zeros((nwin, g.xyzz_limbs), dtype=np.uint64)
    tot[:, :2*L] = pts; tot[:, 2*L:3*L] = one; tot[:, 3*L:] = one
    g.fold_windows(tot, c)
    t0 = time.perf_counter()
    for _ in range(200): g.fold_windows(tot, c)
    print(curve, which, "fold:", round((time.perf_counter()-t0)/200*1e6,1), "us", flush=True)
os.system("grep -m1 'model name' /proc/cpuinfo")
