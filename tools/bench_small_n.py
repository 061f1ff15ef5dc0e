#!/usr/bin/env python3
"""The `small_n` block of bench.py on its own: python tools/bench_small_n.py [curve group] [--no-cpu] [--logns=5,6,...]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    curve, group = (argv + ["bn254", "g1"])[:2]
    kw = {}
    for a in sys.argv[1:]:
        if a.startswith("--logns="):
            kw["logns"] = tuple(int(x) for x in a.split("=", 1)[1].split(","))
    gm = importlib.import_module("gnark-crypto_amd")
    assert gm._lib.load().gmsm_set_device(0) == 0
    torch.cuda.set_device(0)
    out = bench.small_n_block(gm, torch, curve, group, with_cpu="--no-cpu" not in sys.argv, **kw)
    for r in out["rows"]:
        print(" ".join(f"{k}={v}" for k, v in r.items()), file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
