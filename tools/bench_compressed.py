"""The compressed-point entries for every group (gmsm_points_compress / gmsm_points_from_compressed): 2^logn device-made points ->
Bytes() on the device -> decoded again from host bytes, without and with the subgroup step; round trip checked. Wall-clock ms
include the copies over PCIe (the entries take and give host bytes); `rocprofv3 --kernel-trace --stats` over this script gives
the kernels alone (profiles/r06_decompress_stats.md).
usage: python tools/bench_compressed.py [logn] [curve group ...]"""
import importlib
import os
import sys

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
    logn = int(args[0]) if args else 20
    rest = args[1:]
    groups = [tuple(rest[i:i + 2]) for i in range(0, len(rest), 2)] or [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2"),
                                                                         ("bw6_761", "g1"), ("bw6_761", "g2")]
    n = 1 << logn
    stream = torch.cuda.current_stream().cuda_stream
    for curve, group in groups:
        g = (gm.G1Affine if group == "g1" else gm.G2Affine)(curve)
        gj = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
        rng = np.random.default_rng([0x636D70, logn])
        d_a = torch.from_numpy(bench.uniform_scalars(rng, gj, n).view(np.int64)).cuda()
        d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        gj.batch_scalar_mul_device(gj.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
        del d_a
        pts = d_pts.cpu().numpy().view(np.uint64)
        enc_ms = bench.median_ms(lambda: g.Compress(d_points=d_pts.data_ptr(), n=n), reps=3)
        comp, err = g.Compress(d_points=d_pts.data_ptr(), n=n)
        assert err is None
        d_out = torch.zeros_like(d_pts)
        row = {}
        for name, check in (("decode", False), ("decode_subgroup", True)):
            def run(check=check):
                m, err = g.DecodeCompressed(comp, subgroup_check=check, d_out=d_out.data_ptr())
                assert err is None and m == n, err
            row[name] = bench.median_ms(run, reps=3)
        same = bool((d_out.cpu().numpy().view(np.uint64) == pts).all())
        ext = g.coord_limbs // g.curve.fp_limbs
        # products of the base field per decoded point: the exponentiation(s) + x^3, the check, the domain changes (Fp2: the norm's root
        # and one or two roots of t, three base products per Fp2 product)
        chain = bench.sqrt_chain_products(g.curve.p)
        prods = chain + 8 if ext == 1 else int(2.5 * chain) + 30
        print(f"{curve} {group} 2^{logn}: compress {enc_ms:.2f} ms | decode {row['decode']:.2f} ms ({n / row['decode'] / 1e3:.1f} M points/s, "
              f"~{prods} base-field products a point -> {n * prods / row['decode'] / 1e6:.1f} G products/s incl. PCIe) | "
              f"+ subgroup check {row['decode_subgroup']:.2f} ms | round trip {'exact' if same else 'MISMATCH'}", flush=True)
        assert same
        del d_pts, d_out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
