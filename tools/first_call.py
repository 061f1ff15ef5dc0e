"""Cold start of the drop-in entry: what the FIRST MultiExp of a fresh process costs (the reference's first call costs nothing
extra, ecc/bn254/multiexp.go:32).  Run as its own process (bench.py does, N = 1): no torch, no warm context.

    python tools/first_call.py [curve] [group]      ->  one JSON object on stdout

  dlopen_ms          dlopen of libgmsm.so (29 MB: the code objects of six groups are mapped, not yet loaded on the device)
  device_ms          gmsm_device_count + gmsm_set_device: HIP runtime initialisation, the context of device 0
  first_2p10_ms      first gmsm_<curve>_<group>_multiexp of 2^10 points: module load of the code objects, streams, workspaces,
                     pinned buffers, the call itself
  second_2p10_ms     the same call again (what a warm process pays)
  first_2p20_ms      first call of 2^20 points in the same process (buffers grow to size: hipMalloc + pinned staging)
  second_2p20_ms     the same call again
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    group = sys.argv[2] if len(sys.argv) > 2 else "g1"
    path = os.environ.get("GMSM_LIB") or os.path.join(ROOT, "gnark-crypto_amd", "csrc", "libgmsm.so")
    t0 = time.perf_counter()
    L = ctypes.CDLL(path)
    t1 = time.perf_counter()
    L.gmsm_device_count.restype = ctypes.c_int
    ndev = L.gmsm_device_count()
    rc = L.gmsm_set_device(0)
    t2 = time.perf_counter()
    out = {"library_bytes": os.path.getsize(path), "dlopen_ms": (t1 - t0) * 1e3, "device_ms": (t2 - t1) * 1e3, "devices": ndev}
    if ndev < 1 or rc != 0:
        out["error"] = "no device"
        print(json.dumps(out))
        return 1
    # inputs through the package's ctypes mirror (same library handle; not part of what a Go caller pays): on-curve bases
    # from the library's host-side generator, scalars below 2^62 per limb
    os.environ["GMSM_NO_TORCH"] = "1"
    sys.path.insert(0, ROOT)
    import importlib
    import numpy as np
    gm = importlib.import_module("gnark-crypto_amd")
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    al, sl = g.aff_limbs, g.fr_limbs
    n = 1 << 20
    pts = g.generate_points(n, 3, 5)
    sc = np.random.default_rng(1).integers(0, 2**62, size=(n, sl), dtype=np.uint64)
    fn = getattr(L, f"gmsm_{curve}_{group}_multiexp")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    jac = np.zeros(g.jac_limbs, dtype=np.uint64)

    def call(m):
        t = time.perf_counter()
        rc = fn(pts.ctypes.data, m, sc.ctypes.data, m, 0, jac.ctypes.data)
        assert rc == 0, rc
        return (time.perf_counter() - t) * 1e3
    out["first_2p10_ms"] = call(1 << 10)
    out["second_2p10_ms"] = call(1 << 10)
    out["first_2p20_ms"] = call(1 << 20)
    out["second_2p20_ms"] = call(1 << 20)
    out["total_to_first_result_ms"] = out["dlopen_ms"] + out["device_ms"] + out["first_2p10_ms"]
    # another group's first call in the same (now warm) process: what loading ONE group's code object costs - the HIP runtime
    # loads a translation unit's code object when its first kernel is launched (deferred loading), and every group is its own
    # translation unit (gmsm_group_inst.hip), so a BN254-only prover never loads BW6-761's kernels
    other = ("bls12_381", "g1") if (curve, group) != ("bls12_381", "g1") else ("bn254", "g1")
    g2 = gm.G1Jac(other[0])
    pts2 = g2.generate_points(1 << 10, 3, 5)
    sc2 = np.random.default_rng(2).integers(0, 2**62, size=(1 << 10, g2.fr_limbs), dtype=np.uint64)
    fn2 = getattr(L, f"gmsm_{other[0]}_{other[1]}_multiexp")
    fn2.restype = ctypes.c_int
    fn2.argtypes = fn.argtypes
    jac2 = np.zeros(g2.jac_limbs, dtype=np.uint64)
    for key in ("other_group_first_2p10_ms", "other_group_second_2p10_ms"):
        t = time.perf_counter()
        assert fn2(pts2.ctypes.data, 1 << 10, sc2.ctypes.data, 1 << 10, 0, jac2.ctypes.data) == 0
        out[key] = (time.perf_counter() - t) * 1e3
    out["deferred_loading"] = os.environ.get("HIP_ENABLE_DEFERRED_LOADING", "default (1)")
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
