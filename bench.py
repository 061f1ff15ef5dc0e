#!/usr/bin/env python3
"""bench.py -- G1 MSM/sec (BN254) on MI355X, the metric of BASELINE.json.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20]

One "step" = one complete MultiExp (device-resident bases and scalars -> Jacobian result on the host) of
n = 2^logn points.  N = 1: the whole MSM on one GPU (BASELINE config C2 at the default logn = 20).  N > 1 (launched by
torch.distributed.run, one rank per GPU): the windows of the SAME MSM are sharded round-robin over the ranks, every
rank holds all bases, the per-window totals are exchanged with one RCCL all-gather and folded (strong scaling of one
MSM, as BASELINE.json's north_star describes).

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      dominant kernel (k_accumulate): algorithmic bytes per launch (96 B/point x n) / mean
                launch duration from HIP events recorded on the launch stream inside libgmsm; peak = 8000 GB/s
  int_roofline  the bound that actually binds: field multiplications per second against the v_mad_u64_u32 issue peak
  cpu_baseline  the oracle (C restatement of gnark-crypto's algorithm, kind "port") timed on this box's host cores
  bit_exact     GPU affine result == oracle affine result on the timed input
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "wait_prev_group"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the PCIe-inclusive (host buffer) measurements")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two-in-flight (submit/collect) measurement")
    ap.add_argument("--shard", default="auto", choices=["auto", "windows", "points"],
                    help="N>1: decomposition of the one MultiExp over the ranks (gnark-crypto_amd/sharding.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (RCCL process group, device-side all-gather) even with one rank")
    ap.add_argument("--resident", action="store_true",
                    help="register the bases once (gmsm_bases_register) and time MultiExp over the resident form")
    ap.add_argument("--curve", default="bn254", help="exploration only: bn254 | bls12_381 | bw6_761")
    ap.add_argument("--group", default="g1", help="exploration only: g1 | g2")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout. Native libraries write there too (RCCL prints a version banner to the C
    # stdout when the process exits, libdrm complains about amdgpu.ids): keep a private handle on the real stdout for the
    # JSON line and point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks of a node
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(local_rank) == 0, gm._lib.last_error()
    g = (gm.G1Jac if args.group == "g1" else gm.G2Jac)(args.curve)
    n = 1 << args.logn

    # synthetic, deterministic, on-curve inputs: P_i = [k0 + i*k1] G; scalars uniform (stored limbs
    # uniform in [0, r) by rejection)
    rng = np.random.default_rng([0x6D736D, args.logn])
    k0, k1 = int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62))
    pts = g.generate_points(n, k0, k1)
    nl = g.fr_limbs
    sc = np.zeros((n, nl), dtype=np.uint64)
    todo = np.arange(n)
    r_limbs = [(g.curve.r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)]
    top_bits = g.curve.fr_bits - 64 * (nl - 1)
    while todo.size:
        cand = rng.integers(0, 2**64, size=(todo.size, nl), dtype=np.uint64)
        cand[:, nl - 1] &= np.uint64((1 << top_bits) - 1)
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for i in range(nl - 1, -1, -1):
            lt |= eq & (cand[:, i] < np.uint64(r_limbs[i]))
            eq &= cand[:, i] == np.uint64(r_limbs[i])
        sc[todo[lt]] = cand[lt]
        todo = todo[~lt]

    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    c = g.default_window_bits(n)
    nwin = g.num_windows(c)

    sharding = importlib.import_module("gnark-crypto_amd.sharding")
    sharded = world > 1 or args.force_dist
    plan = exchange = None
    if sharded:
        # One MultiExp over all ranks (strong scaling): point or window decomposition (sharding.py), the totals stay on
        # the device until ONE RCCL all-gather; every rank folds.
        plan = sharding.shard_plan(g, n, rank, world, args.shard)
        c, nwin = plan["c"], plan["nwin"]
        exchange = sharding.Exchange(dist, torch.device("cuda", local_rank), plan["rows"], g.xyzz_limbs)
        d_pts_loc, d_sc_loc, n_loc = d_pts[plan["lo"]:plan["hi"]], d_sc[plan["lo"]:plan["hi"]], plan["hi"] - plan["lo"]

    resident = None
    if args.resident:
        resident = (g.register_bases(d_points=d_pts_loc.data_ptr(), n=n_loc) if sharded
                    else g.register_bases(d_points=d_pts.data_ptr(), n=n))

    def enqueue(plan_, local):
        g.window_sums_enqueue(d_pts_loc.data_ptr(), d_sc_loc.data_ptr(), n_loc, plan_["c"], plan_["win_first"],
                              plan_["win_stride"], stream, local.data_ptr(), bases=resident)

    def step():
        if sharded:
            return sharding.sharded_multiexp_exchange(g, plan, enqueue, exchange)
        if resident is not None:
            return resident.multiexp_device(d_sc.data_ptr(), n, stream)
        return g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        jac = step()
    lib.gmsm_set_profiling(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jac = step()
    barrier()
    dt = time.perf_counter() - t0
    stage_ms = (_ctypes_double * len(STAGES))()
    calls = _ctypes_ulong(0)
    lib.gmsm_get_stage_times(stage_ms, len(STAGES), _byref(calls))
    stage_launches = (_ctypes_ulong * len(STAGES))()
    lib.gmsm_get_stage_launches(stage_launches, len(STAGES))
    lib.gmsm_set_profiling(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # Two MultiExp calls in flight over resident bases (gmsm_multiexp_bases_submit/_collect): reported beside the
    # headline number, never instead of it. Every one of the K calls is submitted and collected inside the timed region.
    pipelined = None
    if not sharded and not args.no_pipeline:
        rb2 = resident or g.register_bases(d_points=d_pts.data_ptr(), n=n)

        def run_pipelined(k):
            prev, out = None, None
            for _ in range(k):
                t = rb2.submit(d_sc.data_ptr(), n)
                if prev is not None:
                    out = rb2.collect(prev)
                prev = t
            last = rb2.collect(prev)
            return out if out is not None else last, last

        run_pipelined(max(2, args.warmup))
        barrier()
        tp0 = time.perf_counter()
        jac_a, jac_b = run_pipelined(args.steps)
        barrier()
        dtp = time.perf_counter() - tp0
        pipelined = {"value": round(args.steps / dtp, 3), "unit": "MSM/s", "ms_per_step": round(dtp / args.steps * 1e3, 4),
                     "in_flight": 2, "equal_to_serial_result": bool((g.jac_to_affine(jac_a) == g.jac_to_affine(jac)).all()
                                                    and (g.jac_to_affine(jac_b) == g.jac_to_affine(jac)).all())}
        # the same through the plain blocking entry from two caller threads (what two goroutines calling MultiExp get:
        # BenchmarkManyMultiExpG1Reference, ecc/bn254/multiexp_test.go:385-415); ctypes drops the GIL during the call
        import threading

        def caller(k, out, slot):
            for _ in range(k):
                out[slot] = rb2.multiexp_device(d_sc.data_ptr(), n, stream)

        def run_two_callers(k):
            res = [None, None]
            th = [threading.Thread(target=caller, args=(k - k // 2, res, 0)), threading.Thread(target=caller, args=(k // 2, res, 1))]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            return res
        run_two_callers(4)
        barrier()
        tc0 = time.perf_counter()
        res2 = run_two_callers(args.steps)
        barrier()
        dtc = time.perf_counter() - tc0
        pipelined["two_blocking_callers"] = {
            "value": round(args.steps / dtc, 3), "unit": "MSM/s", "ms_per_step": round(dtc / args.steps * 1e3, 4),
            "equal_to_serial_result": bool(all(r is None or (g.jac_to_affine(r) == g.jac_to_affine(jac)).all() for r in res2))}
        # k MultiExp in ONE blocking call with the scalars on the HOST (gmsm_multiexp_bases_batch): the copy of vector
        # i+1 overlaps the accumulation of vector i, so this rate includes the 32 B/scalar over PCIe
        kb = 8
        sc_batch = np.ascontiguousarray(np.broadcast_to(sc, (kb,) + sc.shape))
        rb2.MultiExpBatch(scalars=sc_batch[:2])
        tb0 = time.perf_counter()
        jb, errb = rb2.MultiExpBatch(scalars=sc_batch)
        dtb = time.perf_counter() - tb0
        assert errb is None
        pipelined["batch_call_host_scalars"] = {
            "value": round(kb / dtb, 3), "unit": "MSM/s", "ms_per_step": round(dtb / kb * 1e3, 4), "k": kb,
            "equal_to_serial_result": bool(all((g.jac_to_affine(j_) == g.jac_to_affine(jac)).all() for j_ in jb))}
        del sc_batch
        if resident is None:
            rb2.release()

    # PCIe-inclusive rates of the drop-in entries (SURVEY.md §8(d) "cold" / "warm-bases"): host buffers in, result out.
    # Reported beside the headline number, never as `value`.
    host_entry = None
    if not sharded and rank == 0 and not args.no_host_entry:
        def median_ms(fn, reps=5):
            fn()
            ts = []
            for _ in range(reps):
                t_ = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t_) * 1e3)
            return sorted(ts)[len(ts) // 2]
        cfg = gm.MultiExpConfig()
        cold = median_ms(lambda: g.MultiExp(pts, sc, cfg))
        rb3 = resident or g.register_bases(d_points=d_pts.data_ptr(), n=n)
        warm = median_ms(lambda: rb3.MultiExp(sc, cfg))
        if resident is None:
            rb3.release()
        host_entry = {"cold_ms": round(cold, 3), "cold_msm_per_s": round(1e3 / cold, 2),
                      "warm_bases_ms": round(warm, 3), "warm_bases_msm_per_s": round(1e3 / warm, 2),
                      "note": "cold: bases+scalars copied from pageable host memory every call (gmsm_<curve>_g1_multiexp); "
                              "warm-bases: registered bases, scalars copied every call (gmsm_multiexp_bases); median of 5"}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.steps / dt
        ncalls = max(1, calls.value)
        stages = {name: stage_ms[i] / ncalls for i, name in enumerate(STAGES)}
        # one MultiExp = several k_accumulate_seg launches (window groups, overlapped with each other's grouping and
        # reduction): per-launch figures are averages over the launches
        acc_launches = max(1, stage_launches[STAGES.index("accumulate")] // ncalls)
        acc_ms = stages["accumulate"] / acc_launches
        bytes_per_point = 8 * (g.aff_limbs + g.fr_limbs)  # affine point + scalar (BN254 G1: 96 B)
        # this rank's share of the (point, window) pairs: all windows of its slice, or its windows of all points
        my_pairs = (plan["hi"] - plan["lo"]) * len(range(plan["win_first"], nwin, plan["win_stride"])) if sharded else n * nwin
        algorithmic_bytes = bytes_per_point * n * my_pairs / (n * nwin) / acc_launches
        achieved = algorithmic_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        # integer roofline of the same kernel: 10 field products per mixed add (8M+2S, g1.go:822), n*(windows of this
        # rank) mixed adds; one lazy 9x29-bit Montgomery product = 171 v_mad_u64_u32/v_mul_lo_u32 + 18 v_lshrrev_b64, all
        # 4 cycles / wave64 / SIMD (tools/ubench_valu.hip): issue peak = 1024 SIMD * 64 lanes * 2.4 GHz / (189 * 4)
        # = 208e9 products/s at the nominal clock; tools/ubench_fpmul.hip measures 174-177e9 on the chip.
        madds = my_pairs / acc_launches
        mulmods_per_s = madds * 10 / (acc_ms * 1e-3) if acc_ms > 0 else 0.0
        int_peak = 1024 * 64 * 2.4e9 / (189 * 4)
        out = {
            "metric": "G1 MSM/sec (BN254)" if (args.curve, args.group) == ("bn254", "g1") else f"{args.group.upper()} MSM/sec ({args.curve})", "value": value, "unit": "MSM/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.curve.upper()} {args.group.upper()} MultiExp 2^{args.logn} points, bases+scalars resident in HBM",
                       "arithmetic": f"{g.curve.p.bit_length()}-bit Montgomery field on 32-bit words (lazy 28/29-bit limbs, v_mad_u64_u32)",
                       "points": n, "window_bits": c, "windows": nwin, "resident_bases": resident is not None,
                       "parallelism": "single GPU" if not sharded else f"{plan['mode']}-sharded x{world} + one RCCL all-gather"},
            "points_per_s": value * n,
            "stage_ms": stages,
            "pipelined": pipelined,
            "host_entry": host_entry,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": measured_traffic(args, world), "kernel": "k_accumulate_seg",
                         "algorithmic_bytes_per_launch": algorithmic_bytes, "avg_launch_ms": acc_ms,
                         "launches_per_msm": acc_launches},
            "int_roofline": {"achieved_mulmod_per_s": mulmods_per_s, "peak_mulmod_per_s": int_peak,
                             "measured_peak_mulmod_per_s": 174e9, "frac": mulmods_per_s / int_peak},
        }
        if sharded:
            # the sharded result against the same MultiExp computed by this rank alone (after the timed region)
            single = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            out["equal_to_single_gpu_result"] = bool((g.jac_to_affine(single) == g.jac_to_affine(jac)).all())
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline(g, pts, sc, jac, args.curve, args.group))
        print(json.dumps(out), file=json_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, world):
    """HBM/fabric bytes per k_accumulate_seg launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
    profiles/traffic_r01.json); counters cannot be read from inside the timed run, so this is the profiled value for
    the same workload, or None when the workload was not profiled."""
    if world != 1 or (args.curve, args.group) != ("bn254", "g1"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            return json.load(f)["k_accumulate_seg"].get(str(args.logn))
    except (OSError, KeyError, ValueError):
        return None


def effective_cpus():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes
    256 hardware threads but limits the container to a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(g, pts, sc, gpu_jac, curve="bn254", group="g1"):
    """The oracle = C restatement of gnark-crypto's MultiExp (bestC, split recursion, one task per (leaf, window),
    extended-Jacobian buckets; the batch-affine bucket variant is off), on all host cores.  Bounded: repeats whole
    MSMs until ~10 s have elapsed (at least 1)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure, used here only as the reported baseline and the checker
    o = oracle.Oracle(curve, group)
    cores = effective_cpus()
    threads = min(2 * cores, len(os.sched_getaffinity(0)))  # 2 software threads per allowed core measured best
    reps, t_total, jac = 0, 0.0, None
    while reps < 1 or (t_total < 10.0 and reps < 50):
        t0 = time.perf_counter()
        err, jac = o.multiexp(pts, sc, nb_tasks=0, num_cpu=cores, nthreads=threads)
        t_total += time.perf_counter() - t0
        reps += 1
        assert err == 0
    exact = bool((o.jac_to_affine(jac) == g.jac_to_affine(gpu_jac)).all())
    return {
        "cpu_baseline": {"value": reps / t_total, "unit": "MSM/s", "cores": cores, "kind": "port",
                         "threads": threads,
                         "sample": f"{reps} full MSM(s) of the same 2^{int(np.log2(len(pts)))} input, {t_total:.1f} s total; "
                                   "C restatement of gnark-crypto's algorithm (ext-Jacobian buckets, batch-affine off)"},
        "bit_exact": exact,
    }


import ctypes as _ct  # noqa: E402

_ctypes_double = _ct.c_double
_ctypes_ulong = _ct.c_ulong
_byref = _ct.byref

if __name__ == "__main__":
    main()
