"""Calls over registered bases with host scalars: the shipped point ranges (unequal for the two narrow G1 groups) against
uniform ranges of the old count (GMSM_OPT_HOST_RANGES forces a count and uniform ranges), plain and with window tables, two rounds
in one session -> profiles/r03_host_skew.log.  usage: python tools/host_skew_ab.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")
stream = torch.cuda.current_stream().cuda_stream
def med(fn, reps=9):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
for curve, which, logn, old in (("bn254","g1",20,2), ("bn254","g1",21,2), ("bn254","g1",22,4), ("bn254","g1",23,8), ("bn254","g1",24,16), ("bls12_381","g1",22,4), ("bn254","g2",20,2), ("bw6_761","g1",20,2)):
    g = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng(7)
    a = rng.integers(0, 2**64, size=(n, g.fr_limbs), dtype=np.uint64); a[:, -1] &= np.uint64((1 << (g.curve.fr_bits - 64 * (g.fr_limbs - 1) - 1)) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    sc = np.roll(a, 1, axis=0)
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
    rbt = None
    if logn <= 21:
        rbt = g.register_bases(d_points=d_pts.data_ptr(), n=n); rbt.precompute(0)
    reps = 9 if logn <= 22 else 5
    res = []
    for rnd in range(2):
        gm.set_option("host_ranges", 0)
        new = med(lambda: rb.MultiExp(sc), reps); newt = med(lambda: rbt.MultiExp(sc), reps) if rbt else 0
        gm.set_option("host_ranges", old)
        o = med(lambda: rb.MultiExp(sc), reps); ot = med(lambda: rbt.MultiExp(sc), reps) if rbt else 0
        res.append(f"new {new:.3f} old({old} uniform) {o:.3f}" + (f" | tables new {newt:.3f} old {ot:.3f}" if rbt else ""))
    gm.set_option("host_ranges", 0)
    print(f"{curve} {which} 2^{logn} warm-bases: " + " ; ".join(res), flush=True)
    rb.release()
    if rbt: rbt.release()
    del d_pts
    torch.cuda.empty_cache()
