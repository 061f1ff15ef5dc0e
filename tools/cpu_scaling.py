"""Thread scaling of the CPU oracle on this host (which thread count gives the best cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
o = oracle.Oracle("bn254", "g1")
N = 1 << 20
P = o.gen_points(N, 12345, 6789, nthreads=64)
S = np.random.default_rng(1).integers(0, 2**62, size=(N, 4), dtype=np.uint64)
for th in (1, 16, 32, 64, 128, 256):
    best = 1e9
    for rep in range(2 if th > 1 else 1):
        t = time.time(); err, jac = o.multiexp(P, S, nb_tasks=0, num_cpu=th, nthreads=th); best = min(best, time.time() - t)
    print("threads", th, "s", round(best, 3))
