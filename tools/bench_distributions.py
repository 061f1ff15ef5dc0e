#!/usr/bin/env python3
"""The `distributions` block of bench.py on its own: python tools/bench_distributions.py [curve group logn steps]... [--no-cold]
(default: BN254 G1 2^20 and 2^24).  One JSON object on stdout, a table on stderr."""
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
    cold = "--no-cold" not in sys.argv
    kinds = None
    for a in sys.argv[1:]:
        if a.startswith("--kinds="):
            kinds = a.split("=", 1)[1].split(",")
    configs = [(argv[i], argv[i + 1], int(argv[i + 2]), int(argv[i + 3])) for i in range(0, len(argv) - 3, 4)]
    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(0) == 0
    for a in sys.argv[1:]:  # --glv=0|1|2: GMSM_OPT_GLV for the whole run
        if a.startswith("--glv="):
            gm.set_option("glv", int(a.split("=", 1)[1]))
    torch.cuda.set_device(0)
    out = bench.distributions_block(gm, lib, torch, *( [tuple(configs)] if configs else []), kinds=kinds, cold=cold)
    for r in out["rows"]:
        st = r["stage_ms"]
        print(f"{r['group']:>14} 2^{r['logn']} {r['distribution']:>12}: {r['ms']:9.3f} ms  x{r['vs_uniform']}  cold {r.get('cold_ms')}  "
              f"bit_exact {r['bit_exact']}  | " + " ".join(f"{k[:4]} {v:.3f}" for k, v in st.items()), file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
