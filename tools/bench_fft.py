"""fr/fft timing on one GPU: python tools/bench_fft.py [curve] [logn ...]"""
import importlib
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gm = importlib.import_module("gnark-crypto_amd")


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    logns = [int(x) for x in sys.argv[2:]] or [16, 20, 22, 24]
    c = gm.CURVES[curve]
    for logn in logns:
        n = 1 << logn
        rng = np.random.default_rng(logn)
        a = rng.integers(0, 2**64, size=(n, c.fr_limbs), dtype=np.uint64)
        a[:, -1] &= np.uint64((1 << (c.fr_bits - 64 * (c.fr_limbs - 1) - 1)) - 1)  # canonical field elements
        t = torch.from_numpy(a.view(np.int64)).cuda()
        d = gm.fft.NewDomain(curve, n)
        stream = torch.cuda.current_stream().cuda_stream
        for dec, coset, name in ((gm.fft.DIF, False, "DIF"), (gm.fft.DIT, False, "DIT"), (gm.fft.DIF, True, "DIF coset")):
            opts = (gm.fft.OnCoset(),) if coset else ()
            d.fft_device(t.data_ptr(), dec, *opts, stream=stream)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                d.fft_device(t.data_ptr(), dec, *opts, stream=stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            bytes_min = 2 * n * 8 * c.fr_limbs  # one read + one write of the vector
            print(f"{curve} fft 2^{logn} {name}: {ms:.3f} ms  ({n * logn / 2 / ms / 1e6:.1f} G butterflies/s, "
                  f"{bytes_min / ms / 1e6:.1f} GB/s of the 2 n x {8 * c.fr_limbs} B minimum)")
        d.release()


if __name__ == "__main__":
    main()
