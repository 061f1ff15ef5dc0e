#!/usr/bin/env python3
"""bench.py -- G1 MSM/sec (BN254) on MI355X, the metric of BASELINE.json.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20]

One "step" = one complete MultiExp (device-resident bases and scalars -> Jacobian result on the host) of
n = 2^logn points.  N = 1: the whole MSM on one GPU (BASELINE config C2 at the default logn = 20).  N > 1 (one rank per GPU under
torch.distributed.run; started as a plain `python bench.py --gpus N` the script re-executes itself under that launcher):
the SAME MSM is sharded over the ranks (points or windows, sharding.py), the per-window totals are exchanged with one
RCCL all-gather and folded (strong scaling of one MSM, as BASELINE.json's north_star describes).

Rank 0 prints one JSON line.  Besides the contract fields it carries
  value_cold / value_warm_bases   SURVEY.md §8(d): the same MSM through the drop-in C entries with host buffers
                (cold: bases + scalars cross PCIe every call; warm-bases: registered bases, scalars cross PCIe)
  roofline      dominant kernel (k_accumulate_seg): algorithmic bytes per launch (96 B/point x n) /
                mean launch duration from HIP events recorded on the launch stream inside libgmsm; peak = 8000 GB/s
  int_roofline  the bound that actually binds: field multiplications per second against the v_mad_u64_u32 issue peak
  cpu_baseline  the oracle (C restatement of gnark-crypto's algorithm, kind "port") timed on this box's host cores
  bit_exact     GPU affine result == oracle affine result on the timed input
  also          the other BASELINE.json configurations, timed in the same run: BN254 G1 2^22, 2^24 (resident,
                warm-bases, cold) and 2^26, BLS12-381 G1 and G2 2^22, BW6-761 G1 2^20; each with ms_per_step, stage_ms,
                roofline (traffic from the committed PMC passes), its own cpu_baseline (the oracle on the same input,
                one repetition) and a closed-form bit_exact check (bases [a_i]G built on the device, expected result
                [sum a_i b_i]G from the oracle: the shape of the reference's own MSM identity, multiexp_test.go:54-60)
  c_abi_sharded the same MultiExp through the drop-in C entry with the library itself spreading it over the devices
                (gmsm_multiexp_sharded / gmsm_bases_register_sharded: one process, one host thread per device)
  replica_batch (--batch K) K MultiExp over the same registered bases spread over the ranks, one all-gather of results
  fft           fr/fft beside the MSM (BN254 2^20 / 2^24, BLS12-381 and BW6-761 fr 2^24, one coset + inverse timing)
  next_rows     SURVEY.md §8(f) N3 / N4-ingest: fixed-base batch 2^20 / 2^24, raw decode + validation 2^22, SRS dump 2^24
  distributions the reference's BenchmarkMultiExpG1 scalar distributions (smallvalues, redundancy) + value_one / all_equal at 2^20 and
                2^24 (BN254 G1) and BLS12-381 G2 2^22: ms, ratio to uniform, stage_ms, cold ms, closed-form bit_exact per row
  small_n       2^5 .. 2^16 points (BN254 G1): resident ms, cold ms through the drop-in entry, the CPU port with one thread
                (NbTasks 1) and with all cores; the measured crossover
  n24, tail     LAST keys of the line (the driver keeps the last 2000 characters): the 2^24 half of the metric in compact
                form, and the headline's value_cold / value_warm_bases / bit_exact
N > 1 adds backend / rccl_ranks / devices_seen (what the process group really was).  --oversubscribe (or
GMSM_BENCH_SHARE_DEVICE=1): the rehearsal of the N > 1 path on fewer devices than ranks - rank r runs on device
r % device_count, the process group is gloo (RCCL refuses two ranks on one device), everything else - shard_plan,
Exchange, sharded_also, host_side_wait, c_abi_sharded, the JSON assembly - is the code the 8-GPU run executes.
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

STAGES = ["decompose", "histogram", "scans", "scatter", "accumulate", "fixup", "reduce", "reserved"]
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md

# BASELINE.json configs beyond the headline one: (curve, group, logn, steps, host-entry legs too)
# (BN254 G1 2^24 - the other half of BASELINE.json's metric - comes LAST: the driver keeps the tail of the line)
ALSO = [("bn254", "g1", 22, 5, True), ("bn254", "g1", 26, 3, False), ("bn254", "g2", 20, 5, False), ("bls12_381", "g1", 22, 5, False),
        ("bls12_381", "g2", 22, 3, False), ("bw6_761", "g1", 20, 3, False), ("bn254", "g1", 24, 5, True)]


def uniform_scalars(rng, g, n):
    """(n, fr_limbs) uint64: stored (Montgomery) limbs uniform in [0, r) by rejection, hence so is the scalar value."""
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
    return sc


STAGE_NOTE = ("accumulate: HIP events around k_accumulate_seg inside the timed region (two events per call); the other stages: "
              "the same steps repeated outside the timed region with every stage bracketed (eight events cost 0.03-0.05 ms per call)")


class StageProfile:
    """Per-stage device times of the calls between start() and stop() (HIP events inside libgmsm)."""

    def __init__(self, lib, level=1):
        """level 1: every stage (eight events per pipeline run, 0.03-0.05 ms per call); level 2: only the accumulation
        kernel is bracketed - what the timed region runs with, so that `value` carries two events per call and the
        roofline's kernel duration is still measured live inside it."""
        self.lib, self.level = lib, level

    def start(self):
        self.lib.gmsm_set_profiling(self.level)

    def stop(self):
        ms = (_ct.c_double * len(STAGES))()
        calls = _ct.c_ulong(0)
        launches = (_ct.c_ulong * len(STAGES))()
        self.lib.gmsm_get_stage_times(ms, len(STAGES), _ct.byref(calls))
        self.lib.gmsm_get_stage_launches(launches, len(STAGES))
        self.lib.gmsm_set_profiling(0)
        ncalls = max(1, calls.value)
        stages = {name: ms[i] / ncalls for i, name in enumerate(STAGES)}
        acc_launches = max(1, launches[STAGES.index("accumulate")] // ncalls)
        return stages, acc_launches


def roofline_record(g, n, pairs_share, stages, acc_launches, traffic):
    """roofline + int_roofline of k_accumulate_seg.  pairs_share = this rank's fraction of the n*nwin (point, window)
    pairs; one MultiExp may run several launches (window groups): per-launch figures are averages."""
    acc_ms = stages["accumulate"] / acc_launches
    bytes_per_point = 8 * (g.aff_limbs + g.fr_limbs)  # affine point + scalar (BN254 G1: 96 B)
    algorithmic_bytes = bytes_per_point * n * pairs_share / acc_launches
    achieved = algorithmic_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "kernel": "k_accumulate_seg", "algorithmic_bytes_per_launch": algorithmic_bytes,
            "avg_launch_ms": acc_ms, "launches_per_msm": acc_launches}


def product_peak(g):
    """(measured products/s, key) of the group's coordinate field: the LARGER of the signed (accumulation loops) and unsigned
    product routines of tools/ubench_peaks.hip (profiles/peaks_r06.json) - the yardstick of int_roofline.frac_of_measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "peaks_r06.json")) as f:
            pk = json.load(f)
    except (OSError, ValueError):
        return None, None
    field = "fp2" if g.coord_limbs != g.curve.fp_limbs else "fp"
    keys = [f"{g.curve.name}_{field}_mul", f"{g.curve.name}_{field}_mul_unsigned"]
    best = max((k for k in keys if k in pk), key=lambda k: pk[k], default=None)
    return (pk[best], best) if best else (None, None)


def int_roofline_record(g, madds, acc_ms_total):
    """Integer roofline of the accumulation kernel: 10 products of the coordinate field per mixed addition (8M + 2S, g1.go:822;
    over Fp2 a product is one lz/f2s product = 4 base-field product scans with two reductions) against (a) the sustained rate of
    the product routine itself on this chip (product_peak) and (b), BN254 G1 only, the nominal-clock issue bound: one 9x29-bit
    product = 171 v_mad_u64_u32 / v_mul_lo_u32 + 18 v_lshrrev_b64 at 4 cycles / wave64 / SIMD (tools/ubench_valu.hip):
    1024 SIMDs * 64 lanes * 2.4 GHz / (189 * 4) = 208e9 products/s."""
    rate = madds * 10 / (acc_ms_total * 1e-3) if acc_ms_total > 0 else 0.0
    measured, key = product_peak(g)
    nominal = 1024 * 64 * 2.4e9 / (189 * 4) if (g.curve.name == "bn254" and g.coord_limbs == g.curve.fp_limbs) else None
    return {"achieved_mulmod_per_s": rate, "peak_mulmod_per_s": nominal, "measured_peak_mulmod_per_s": measured, "measured_peak_routine": key,
            "frac": (rate / nominal) if nominal else None, "frac_of_measured": (rate / measured) if measured else None}


def measured_traffic(curve, group, logn, world, nwin=None):
    """HBM/fabric bytes per k_accumulate_seg launch from the committed rocprofv3 PMC passes of THIS round's build
    (FETCH_SIZE + WRITE_SIZE, separate passes; profiles/traffic_r05.json, keyed curve_group_logn).  Counters cannot be
    read from inside the timed run, so this is the profiled value for the same workload - or None when that workload was
    not profiled."""
    if world != 1:
        return None
    try:
        path = next(p for p in (os.path.join(ROOT, "profiles", f"traffic_r0{r}.json") for r in (6, 5, 4)) if os.path.exists(p))
        with open(path) as f:
            rec = json.load(f)
        key = f"{curve}_{group}_{logn}"
        if nwin is not None and rec.get("windows", {}).get(key, nwin) != nwin:
            return None  # profiled with another window width: not this workload's traffic
        return rec["k_accumulate_seg"].get(key)
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def median_ms(fn, reps=5, warm=1):
    """median wall-clock ms of `reps` calls of a BLOCKING entry (host buffers in, result out), after `warm` untimed ones"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t_ = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t_) * 1e3)
    return sorted(ts)[len(ts) // 2]


def loop_ms(fn, reps, sync=None, warm=1):
    """(mean ms, last result) of `reps` back-to-back calls between two `sync()`s, after `warm` untimed calls - the one timing
    loop of every row beside the headline's own (main() keeps its loop in the open: barrier, K steps, barrier)"""
    out = None
    for _ in range(warm):
        out = fn()
    if sync is not None:
        sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    if sync is not None:
        sync()
    return (time.perf_counter() - t0) / reps * 1e3, out


def tables_record(gm, g, d_pts, d_sc, sc, n, stream, jac, steps):
    """MultiExp over registered bases with window tables: ms with the scalars resident / in host memory / two tickets in
    flight, against the same handle without tables (GMSM_OPT_TABLES = 0), result compared with the headline call's."""
    import torch
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
    try:
        t0 = time.perf_counter()
        c = rb.precompute(0)
        build_ms = (time.perf_counter() - t0) * 1e3
        cfg = gm.MultiExpConfig()

        def resident_ms():
            return loop_ms(lambda: rb.multiexp_device(d_sc.data_ptr(), n, stream), steps, torch.cuda.synchronize)

        def in_flight_ms():
            def run(k):
                prev = None
                for _ in range(k):
                    t = rb.submit(d_sc.data_ptr(), n)
                    if prev is not None:
                        rb.collect(prev)
                    prev = t
                return rb.collect(prev)
            run(3)
            ms, out = loop_ms(lambda: run(steps), 1, warm=0)
            return ms / steps, out

        runs0 = int(gm._lib.load().gmsm_debug_table_runs())
        dev_ms, j1 = resident_ms()
        used = int(gm._lib.load().gmsm_debug_table_runs()) > runs0
        host_ms = median_ms(lambda: rb.MultiExp(sc, cfg))
        fl_ms, j2 = in_flight_ms()
        with gm.options(tables=0):
            plain_ms, _ = resident_ms()
            plain_host_ms = median_ms(lambda: rb.MultiExp(sc, cfg))
        ref = g.jac_to_affine(jac)
        return {"window_bits": c, "slabs": g.num_windows(c), "table_bytes": g.num_windows(c) * n * g.aff_limbs * 8,
                "build_ms": round(build_ms, 1), "through_tables": used,
                "device_scalars_ms": round(dev_ms, 4), "device_scalars_msm_per_s": round(1e3 / dev_ms, 2),
                "host_scalars_ms": round(host_ms, 3), "host_scalars_msm_per_s": round(1e3 / host_ms, 2),
                "two_in_flight_ms": round(fl_ms, 4), "two_in_flight_msm_per_s": round(1e3 / fl_ms, 2),
                "same_handle_without_tables": {"device_scalars_ms": round(plain_ms, 4), "host_scalars_ms": round(plain_host_ms, 3)},
                "equal_to_headline_result": bool((g.jac_to_affine(j1) == ref).all() and (g.jac_to_affine(j2) == ref).all()),
                "note": "registered bases + gmsm_bases_precompute: 2^(c w) P_i for every window in HBM, all windows share one "
                        "bucket set and one reduction; the reference has no counterpart (it takes the bases anew per call)"}
    finally:
        rb.release()


def also_config(gm, lib, torch, curve, group, logn, steps, host_legs, with_cpu=True):
    """One of the other BASELINE.json configurations on this GPU: bases [a_i]G built ON the device (fixed-base batch,
    gmsm_batch_scalar_mul_device), uniform scalars b_i, K timed MultiExp calls over resident inputs, and the closed form
    [sum a_i b_i]G from the oracle as the bit-exactness check."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure: here only the checker
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng([0x6D736D, logn, len(curve), 1 if group == "g2" else 0])
    a = uniform_scalars(rng, g, n)
    b = uniform_scalars(rng, g, n)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_b = torch.from_numpy(b.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    del d_a
    plan = g.default_plan(n)  # width, windows, GLV (2 entries per point) of the call as the library runs it
    c, nwin = plan["window_bits"], plan["windows"]
    jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)  # warm-up (allocations, LDS attributes)
    prof = StageProfile(lib, level=2)
    prof.start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        jac = g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc_only, acc_launches = prof.stop()
    prof = StageProfile(lib)  # the stage breakdown: the same steps once more, outside the timed region
    prof.start()
    for _ in range(steps):
        g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
    torch.cuda.synchronize()
    stages, _ = prof.stop()
    stages["accumulate"] = acc_only["accumulate"]  # the roofline's kernel duration is the timed region's
    out = {"workload": f"{curve.upper()} {group.upper()} MultiExp 2^{logn} points, bases+scalars resident in HBM",
           "value": steps / dt, "unit": "MSM/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "window_bits": c,
           "windows": nwin, "glv": plan["entries_per_point"] == 2, "stage_ms": stages, "stage_ms_note": STAGE_NOTE,
           "roofline": roofline_record(g, n, 1.0, stages, acc_launches, measured_traffic(curve, group, logn, 1, nwin)),
           "int_roofline": int_roofline_record(g, n * nwin * plan["entries_per_point"], stages["accumulate"])}
    pts_host = d_pts.cpu().numpy().view(np.uint64) if (host_legs or with_cpu) else None
    if host_legs:
        cfg = gm.MultiExpConfig()
        rb = g.register_bases(d_points=d_pts.data_ptr(), n=n)
        warm = median_ms(lambda: rb.MultiExp(b, cfg), reps=3)
        jw, _ = rb.MultiExp(b, cfg)
        rb.release()
        cold = median_ms(lambda: g.MultiExp(pts_host, b, cfg), reps=3)
        jc, _ = g.MultiExp(pts_host, b, cfg)
        out.update({"value_warm_bases": 1e3 / warm, "warm_bases_ms": warm, "value_cold": 1e3 / cold, "cold_ms": cold,
                    "host_entries_equal_resident": bool((g.jac_to_affine(jw) == g.jac_to_affine(jac)).all()
                                                        and (g.jac_to_affine(jc) == g.jac_to_affine(jac)).all())})
    if with_cpu:  # the CPU port on the same input, one repetition (the baseline beside every timed size)
        rec = cpu_baseline(g, pts_host, b, jac, curve, group, min_seconds=0.0)
        out["cpu_baseline"] = rec["cpu_baseline"]
        out["bit_exact_vs_cpu_port"] = rec["bit_exact"]
    del pts_host
    t0 = time.perf_counter()
    expected = oracle.Oracle(curve, group).fixed_base_msm_affine(a, b)
    out["bit_exact"] = bool((g.jac_to_affine(jac) == expected).all())
    out["bit_exact_check"] = f"closed form [sum a_i b_i]G via the oracle ({time.perf_counter() - t0:.1f} s on one host core)"
    del d_pts, d_b
    torch.cuda.empty_cache()
    return out


# The reference's own benchmark matrix (BenchmarkMultiExpG1, ecc/bn254/multiexp_test.go:301-364) times every size under three
# scalar distributions; provers add vectors full of 0 / 1 / repeated values. Stored (Montgomery) limbs, as the reference builds them.
DISTRIBUTIONS = ["uniform", "smallvalues", "redundancy", "value_one", "all_equal"]


def skewed_scalars(kind, base, g, rng):
    """The scalar vector of one distribution, derived from the uniform vector `base` (n, fr_limbs):
      smallvalues  every 5th scalar SetZero(); [0] = 1 (multiexp_test.go:319-325): the STORED limbs are 1, i.e. the value
                   R^-1 mod r - the same full-width scalar n/5 times, one crowded bucket in EVERY window
      redundancy   runs of 100 equal scalars (:327-334)
      value_one    30 % of the scalars are the field element 1 (Montgomery R mod r): one crowded bucket in window 0 only
      all_equal    every scalar the same"""
    n = base.shape[0]
    sc = base.copy()
    if kind == "smallvalues":
        sc[::5] = 0
        sc[::5, 0] = 1
    elif kind == "redundancy":
        sc = np.ascontiguousarray(np.repeat(base[: (n + 99) // 100], 100, axis=0)[:n])
    elif kind == "value_one":
        one = (1 << (64 * g.fr_limbs)) % g.curve.r
        sc[rng.random(n) < 0.3] = np.array([(one >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(g.fr_limbs)], dtype=np.uint64)
    elif kind == "all_equal":
        sc = np.ascontiguousarray(np.tile(base[:1], (n, 1)))
    elif kind != "uniform":
        raise ValueError(kind)
    return sc


def distributions_block(gm, lib, torch, configs=(("bn254", "g1", 20, 10), ("bn254", "g1", 24, 3)), kinds=None, cold=True):
    """MultiExp under the reference's benchmark distributions: per (group, size) one row per distribution with the resident
    ms (K timed calls), its ratio to the uniform row, the stage breakdown, the cold host-entry ms and a closed-form
    bit_exact (bases [a_i]G built on the device, expected [sum a_i b_i]G from the oracle - valid for any b)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure: here only the checker
    rows = []
    for curve, group, logn, steps in configs:
        g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
        n = 1 << logn
        rng = np.random.default_rng([0x646973, logn, len(curve), 1 if group == "g2" else 0])
        a = uniform_scalars(rng, g, n)
        base = uniform_scalars(rng, g, n)
        d_a = torch.from_numpy(a.view(np.int64)).cuda()
        d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
        del d_a
        pts_host = d_pts.cpu().numpy().view(np.uint64) if cold else None
        o = oracle.Oracle(curve, group)
        uniform_ms = None
        for kind in (kinds or DISTRIBUTIONS):
            b = skewed_scalars(kind, base, g, rng)
            d_b = torch.from_numpy(b.view(np.int64)).cuda()
            ms, jac = loop_ms(lambda: g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream), steps, torch.cuda.synchronize)
            prof = StageProfile(lib)
            prof.start()
            for _ in range(max(2, steps // 2)):
                g.multiexp_device(d_pts.data_ptr(), d_b.data_ptr(), n, stream)
            torch.cuda.synchronize()
            stages, _ = prof.stop()
            if kind == "uniform":
                uniform_ms = ms
            row = {"group": f"{curve}_{group}", "logn": logn, "distribution": kind, "ms": round(ms, 4),
                   "vs_uniform": round(ms / uniform_ms, 3) if uniform_ms else None,
                   "stage_ms": {k: round(v, 4) for k, v in stages.items() if k != "reserved"}}
            if cold:
                cfg = gm.MultiExpConfig()
                row["cold_ms"] = round(median_ms(lambda: g.MultiExp(pts_host, b, cfg), reps=3), 3)
            row["bit_exact"] = bool((g.jac_to_affine(jac) == o.fixed_base_msm_affine(a, b)).all())
            rows.append(row)
            del d_b
        del d_pts, pts_host
        torch.cuda.empty_cache()
    return {"rows": rows, "worst_vs_uniform": max((r["vs_uniform"] or 0.0) for r in rows),
            "all_bit_exact": all(r["bit_exact"] for r in rows),
            "note": "the reference's BenchmarkMultiExpG1 distributions (multiexp_test.go:301-364) + value_one / all_equal; "
                    "ms = resident inputs, cold_ms = host buffers through the drop-in entry (median of 3); bit_exact = closed form"}


def small_n_block(gm, torch, curve="bn254", group="g1", logns=(2, 3, 4, 5, 6, 8, 10, 12, 14, 16), reps=30, with_cpu=True):
    """Sizes below 2^20 (the reference benches from 2^5, multiexp_test.go:344; Pedersen commits with NbTasks: 1,
    fr/pedersen/pedersen.go:100-131): per size the resident ms, the cold drop-in entry (host buffers in, Jacobian out), and
    the CPU port with one thread (NbTasks 1) and with all cores - the measured GPU/CPU crossover the Go stub routes by."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure: the reported CPU legs and the checker
    g = (gm.G1Jac if group == "g1" else gm.G2Jac)(curve)
    o = oracle.Oracle(curve, group)
    nmax = 1 << max(logns)
    rng = np.random.default_rng([0x736D6C, len(curve), 1 if group == "g2" else 0])
    pts = g.generate_points(nmax, int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62)))
    sc = uniform_scalars(rng, g, nmax)
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    cfg = gm.MultiExpConfig()
    cores = effective_cpus()
    rows = []
    rb = g.register_bases(d_points=d_pts.data_ptr(), n=nmax)  # registered bases with window tables (a resident SRS): the third column
    rb.precompute(0)
    for logn in logns:
        n = 1 << logn
        res_ms, jac = loop_ms(lambda: g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream), reps, torch.cuda.synchronize)
        p_, s_ = np.ascontiguousarray(pts[:n]), np.ascontiguousarray(sc[:n])
        cold_ms = median_ms(lambda: g.MultiExp(p_, s_, cfg), reps=9)
        jc, _ = g.MultiExp(p_, s_, cfg)
        tab_ms, jt = loop_ms(lambda: rb.multiexp_device(d_sc.data_ptr(), n, stream), reps, torch.cuda.synchronize)
        row = {"logn": logn, "resident_ms": round(res_ms, 4), "cold_ms": round(cold_ms, 4), "registered_tables_ms": round(tab_ms, 4)}
        expected = o.msm_affine(p_, s_, nthreads=2 * cores)
        row["bit_exact"] = bool((g.jac_to_affine(jac) == expected).all() and (g.jac_to_affine(jc) == expected).all()
                                and (g.jac_to_affine(jt) == expected).all())
        if with_cpu:
            def cpu_ms(nb_tasks, threads):
                k, t_tot = 0, 0.0
                while k < 3 or (t_tot < 0.2 and k < 200):
                    t1 = time.perf_counter()
                    o.multiexp(p_, s_, nb_tasks=nb_tasks, num_cpu=cores if nb_tasks == 0 else 1, nthreads=threads)
                    t_tot += time.perf_counter() - t1
                    k += 1
                return t_tot / k * 1e3
            row["cpu_port_1thread_ms"] = round(cpu_ms(1, 1), 4)
            row["cpu_port_allcores_ms"] = round(cpu_ms(0, min(2 * cores, len(os.sched_getaffinity(0)))), 4)
            row["gpu_cold_over_cpu_allcores"] = round(cold_ms / row["cpu_port_allcores_ms"], 3)
        rows.append(row)
    rb.release()
    cross = next((r["logn"] for r in rows if with_cpu and r["cold_ms"] < min(r["cpu_port_1thread_ms"], r["cpu_port_allcores_ms"])), None)
    return {"group": f"{curve}_{group}", "rows": rows, "cpu_cores": cores, "cpu_kind": "port",
            "crossover_logn_cold_vs_cpu_port": cross,
            "note": "resident: bases+scalars in HBM; cold: gmsm_<curve>_<group>_multiexp with host buffers (median of 9); registered_tables: "
                    "gmsm_multiexp_bases_device over registered bases with window tables (narrow tables up to 2^12 points: one total, no host fold); CPU: the "
                    "oracle's restatement of the reference's MultiExp with NbTasks 1 / all cores on this box"}


def host_side_wait(dist, rank, key, work):
    """Rank 0 runs work() while the other ranks wait on the HOST (a key in the rendezvous store): an RCCL barrier would
    keep their GPUs spinning in a collective kernel, and rank 0 is about to use those GPUs from its own process."""
    store = dist.distributed_c10d._get_default_store()
    out = None
    if rank == 0:
        try:
            out = work()
        finally:
            store.set(key, "1")
    else:
        store.wait([key])
    return out


def c_abi_sharded(gm, g, pts, sc, devices, reference_affine, reps=5):
    """The drop-in C entry with the LIBRARY spreading the MultiExp over `devices` (one process, one host thread per
    logical rank, window totals back through each device's pinned buffer, one fold): cold = bases + scalars from host
    memory every call (every device pulls its slice over its own PCIe link), warm-bases = gmsm_bases_register_sharded
    once, scalars from host memory every call."""
    cfg = gm.MultiExpConfig()
    cold = median_ms(lambda: g.MultiExpSharded(pts, sc, cfg, devices=devices), reps=reps)
    jc, err = g.MultiExpSharded(pts, sc, cfg, devices=devices)
    assert err is None, err
    t0 = time.perf_counter()
    rbs = g.register_bases_sharded(pts, devices=devices)
    t_reg = (time.perf_counter() - t0) * 1e3
    warm = median_ms(lambda: rbs.MultiExp(sc, cfg), reps=reps)
    jw, err = rbs.MultiExp(sc, cfg)
    assert err is None, err
    rbs.release()
    return {"entry": "gmsm_multiexp_sharded / gmsm_bases_register_sharded + gmsm_multiexp_bases (host buffers, one process)",
            "devices": list(devices), "points": int(pts.shape[0]), "cold_ms": round(cold, 3), "value_cold": round(1e3 / cold, 2),
            "warm_bases_ms": round(warm, 3), "value_warm_bases": round(1e3 / warm, 2), "register_ms": round(t_reg, 1),
            "unit": "MSM/s", "equal_to_reference_result": bool((g.jac_to_affine(jc) == reference_affine).all()
                                                               and (g.jac_to_affine(jw) == reference_affine).all())}


def dist_max(torch, dist, seconds):
    """max over the ranks of a host-side duration (the tensor lives where the process group's backend wants it)."""
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


SHARD_CHUNK_LOG = 20  # synthetic inputs of the sharded rows are generated in chunks of 2^20, seeded per chunk: a rank builds only what it owns


def chunked_scalars(g, logn, tag, lo, hi):
    """Rows [lo, hi) of the (2^logn, fr_limbs) uniform scalar array `tag` (0: the a_i of the bases [a_i]G, 1: the b_i), the
    same on every rank whatever its slice: chunk k is seeded by (logn, tag, k)."""
    ch = 1 << min(SHARD_CHUNK_LOG, logn)
    out = np.empty((hi - lo, g.fr_limbs), dtype=np.uint64)
    for k in range(lo // ch, (hi + ch - 1) // ch):
        rows = uniform_scalars(np.random.default_rng([0x6D736D, logn, 5, tag, k]), g, ch)
        s0, s1 = max(lo, k * ch), min(hi, (k + 1) * ch)
        out[s0 - lo:s1 - lo] = rows[s0 - k * ch:s1 - k * ch]
    return out


def max_over_ranks(dist, world, value):
    """max over the ranks of a dict of floats (or a float), through the process group's object gather."""
    if world == 1:
        return value
    got = [None] * world
    dist.all_gather_object(got, value)
    if isinstance(value, dict):
        return {k: max(v[k] for v in got) for k in value}
    return max(got)


def exchange_breakdown(torch, g, sharding, plan, enqueue, exchange, reps=3):
    """One sharded MultiExp cut at its two joints (a synchronize after the local pipeline, one after the gather): ms of
    the local compute, of the exchange (all-gather + copy of the gathered block to the host) and of the fold. A breakdown,
    measured outside the timed loop - the timed loop runs the three back to back."""
    comp = gath = fold = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(plan, exchange.local)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gathered = exchange.gather()
        t2 = time.perf_counter()
        if plan["mode"] == "points":
            g.fold_window_sets(gathered, plan["c"])
        else:
            g.fold_windows(sharding.unpack_gathered(gathered, plan["nwin"], exchange.world, g.xyzz_limbs), plan["c"])
        t3 = time.perf_counter()
        comp, gath, fold = comp + (t1 - t0), gath + (t2 - t1), fold + (t3 - t2)
    return {"compute_ms": comp / reps * 1e3, "gather_ms": gath / reps * 1e3, "fold_ms": fold / reps * 1e3}


def sharded_also(gm, lib, torch, dist, sharding, rank, world, dev_index, mode, rank_devices, logn=24, steps=5, with_c_abi=True):
    """BASELINE.json configs[2]: BN254 G1 2^logn as ONE MultiExp over all ranks, timed like the headline loop (barrier +
    synchronize on both sides, max over ranks). Every rank builds ITS bases [a_i]G on its device from per-chunk seeds; the
    result is checked against the closed form [sum a_i b_i]G (every rank contributes the dot product of its point slice).
    stage_ms = max over ranks of the per-stage device times, exchange_ms = all-gather + D2H + fold (max over ranks)."""
    t_leg = time.perf_counter()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure: the checker
    g = gm.G1Jac("bn254")
    n = 1 << logn
    stream = torch.cuda.current_stream().cuda_stream
    plan = sharding.shard_plan(g, n, rank, world, mode)
    lo, hi = plan["lo"], plan["hi"]
    a = chunked_scalars(g, logn, 0, lo, hi)
    b = chunked_scalars(g, logn, 1, lo, hi)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_b = torch.from_numpy(b.view(np.int64)).cuda()
    d_pts = torch.empty((hi - lo, g.aff_limbs), dtype=torch.int64, device="cuda")
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), hi - lo, d_pts.data_ptr(), stream)
    del d_a
    exchange = sharding.Exchange(dist, torch.device("cuda", dev_index), plan["rows"], g.xyzz_limbs)

    def enqueue(plan_, local):
        g.window_sums_enqueue(d_pts.data_ptr(), d_b.data_ptr(), hi - lo, plan_["c"], plan_["win_first"], plan_["win_stride"],
                              stream, local.data_ptr())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    jac = sharding.sharded_multiexp_exchange(g, plan, enqueue, exchange)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        jac = sharding.sharded_multiexp_exchange(g, plan, enqueue, exchange)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = dist_max(torch, dist, dt)
    prof = StageProfile(lib)  # the stage breakdown: the same steps once more, outside the timed region
    prof.start()
    for _ in range(steps):
        sharding.sharded_multiexp_exchange(g, plan, enqueue, exchange)
    sync()
    stages, _ = prof.stop()
    stages = max_over_ranks(dist, world, {k: v for k, v in stages.items() if k != "reserved"})
    brk = max_over_ranks(dist, world, exchange_breakdown(torch, g, sharding, plan, enqueue, exchange))
    sync()
    # closed form: sum a_i b_i over this rank's share of the points (window mode: every rank holds them all)
    fr = oracle.Field("bn254_fr", g.fr_limbs)
    s_lo, s_hi = sharding.point_slice(n, rank, world) if plan["mode"] == "windows" else (lo, hi)
    part = fr.from_mont(fr.dot(a[s_lo - lo:s_hi - lo], b[s_lo - lo:s_hi - lo]))
    part = sum(int(v) << (64 * i) for i, v in enumerate(part))
    parts = [part]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, part)
    out = {"workload": f"BN254 G1 MultiExp 2^{logn} points, {plan['mode']}-sharded x{world} + one all-gather ({dist.get_backend()})",
           "value": steps / dt, "unit": "MSM/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "n_gpus": world,
           "scaling": "strong", "window_bits": plan["c"], "stage_ms": {k: round(v, 4) for k, v in stages.items()},
           "stage_ms_note": "max over the ranks of each stage's device time, measured outside the timed loop",
           "compute_ms": round(brk["compute_ms"], 4), "exchange_ms": round(brk["gather_ms"] + brk["fold_ms"], 4),
           "exchange_parts_ms": {"all_gather_and_d2h": round(brk["gather_ms"], 4), "fold": round(brk["fold_ms"], 4)}}
    expected = None
    if rank == 0:
        o = oracle.Oracle("bn254", "g1")
        expected = o.jac_to_affine(o.scalar_mul(o.generator, sum(parts) % g.curve.r))
        out["bit_exact"] = bool((g.jac_to_affine(jac) == expected).all())
    del d_pts, d_b
    torch.cuda.empty_cache()

    def through_the_c_abi():  # rank 0 alone drives all `world` devices from its one process
        a0 = chunked_scalars(g, logn, 0, 0, n)
        b0 = chunked_scalars(g, logn, 1, 0, n)
        d_a0 = torch.from_numpy(a0.view(np.int64)).cuda()
        d_p0 = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        g.batch_scalar_mul_device(g.generator, d_a0.data_ptr(), n, d_p0.data_ptr(), stream)
        pts_host = d_p0.cpu().numpy().view(np.uint64)
        del d_a0, d_p0
        torch.cuda.empty_cache()
        return c_abi_sharded(gm, g, pts_host, b0, rank_devices, expected, reps=3)
    if world > 1 and with_c_abi:
        rec = host_side_wait(dist, rank, f"c_abi_{logn}", through_the_c_abi)
        if rank == 0:
            out["c_abi_sharded"] = rec
    out["leg_elapsed_s"] = round(time.perf_counter() - t_leg, 1)
    print(f"[bench] sharded 2^{logn} x{world}: {out['ms_per_step']:.3f} ms/step, leg {out['leg_elapsed_s']} s", file=sys.stderr, flush=True)
    return out


def sqrt_chain_products(q):
    """squarings + products of gmsm_decompress.h's x^(q >> 2): sliding window of three bits over x, x^3, x^5, x^7"""
    bit = lambda j: (q >> (j + 2)) & 1
    j, ops, started = q.bit_length() - 3, 4, False  # x^2 and the three odd powers
    while j >= 0:
        if not bit(j):
            ops, j = ops + 1, j - 1
            continue
        lo = max(j - 2, 0)
        while not bit(lo):
            lo += 1
        ops += (j - lo + 2) if started else 0  # the window's squarings and its product
        started, j = True, lo - 1
    return ops


def fft_config(gm, torch, curve="bn254", logn=24, reps=5):
    """fr/fft next to the MSM: (*Domain).FFT DIF on 2^logn resident coefficients, and the round trip
    FFTInverse(DIT) o FFT(DIF) == identity as the size-independent check (fft_test.go:160-180)."""
    c = gm.CURVES[curve]
    n = 1 << logn
    rng = np.random.default_rng([0x666674, logn])
    a = rng.integers(0, 2**64, size=(n, c.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (c.fr_bits - 64 * (c.fr_limbs - 1) - 1)) - 1)  # below 2^(fr_bits-1) < r: canonical elements
    t = torch.from_numpy(a.view(np.int64)).cuda()
    d = gm.fft.NewDomain(curve, n)
    stream = torch.cuda.current_stream().cuda_stream
    d.fft_device(t.data_ptr(), gm.fft.DIF, stream=stream)
    d.fft_device(t.data_ptr(), gm.fft.DIT, inverse=True, stream=stream)
    ok = bool((t.cpu().numpy().view(np.uint64) == a).all())
    ms, _ = loop_ms(lambda: d.fft_device(t.data_ptr(), gm.fft.DIF, stream=stream), reps, torch.cuda.synchronize, warm=0)
    d.release()
    passes = 1 + (max(0, logn - 10) + 7) // 8  # gmsm_fft.h: the low 10 bits in one pass, the rest in passes of <= 8
    traffic = passes * 2 * n * 8 * c.fr_limbs
    return {"workload": f"{curve.upper()} fr FFT (DIF) 2^{logn} elements resident in HBM", "ms": ms, "ffts_per_s": 1e3 / ms,
            "butterflies_per_s": n * logn / 2 / (ms * 1e-3), "hbm_passes": passes,
            "roofline": {"bound": "hbm", "achieved": traffic / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "note": "compulsory bytes of the passes / wall time; the kernel is VALU-issue-bound (profiles/r04_fft_stats.md: "
                                 "SQ_ACTIVE_INST_VALU 96 % of its cycles, ~370 instructions per butterfly, 190 of them the lazy-limb product)"},
            "round_trip_exact": ok}


def fft_extra(gm, torch, curve, logn=24, reps=3):
    """One coset transform and its inverse (FFT(DIF, OnCoset) then FFTInverse(DIT, OnCoset), fft.go:31-196) on 2^logn resident
    elements: both timings and the exact round trip."""
    c = gm.CURVES[curve]
    n = 1 << logn
    rng = np.random.default_rng([0x666674, logn, 1])
    a = rng.integers(0, 2**64, size=(n, c.fr_limbs), dtype=np.uint64)
    a[:, -1] &= np.uint64((1 << (c.fr_bits - 64 * (c.fr_limbs - 1) - 1)) - 1)
    t = torch.from_numpy(a.view(np.int64)).cuda()
    d = gm.fft.NewDomain(curve, n)
    stream = torch.cuda.current_stream().cuda_stream
    coset = gm.fft.OnCoset()
    d.fft_device(t.data_ptr(), gm.fft.DIF, coset, stream=stream)  # first use builds the coset tables
    d.fft_device(t.data_ptr(), gm.fft.DIT, coset, inverse=True, stream=stream)
    ok = bool((t.cpu().numpy().view(np.uint64) == a).all())
    fwd = inv = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.fft_device(t.data_ptr(), gm.fft.DIF, coset, stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d.fft_device(t.data_ptr(), gm.fft.DIT, coset, inverse=True, stream=stream)
        torch.cuda.synchronize()
        fwd += (t1 - t0) * 1e3 / reps
        inv += (time.perf_counter() - t1) * 1e3 / reps
    d.release()
    return {"workload": f"{curve.upper()} fr coset FFT (DIF) + coset FFTInverse (DIT), 2^{logn} elements resident in HBM",
            "coset_fft_ms": fwd, "coset_inverse_ms": inv, "round_trip_exact": ok}


MEASURED_MULMOD_PER_S = 174e9  # bare 9x29-bit lazy product on this chip (tools/ubench_fpmul.hip, profiles/peaks_r02.json)


def next_rows(gm, lib, torch):
    """SURVEY.md §8(f) N3 / N4-ingest next to the MSM, BN254 G1: the fixed-base batch that builds an SRS
    (BatchScalarMultiplicationG1, ecc/bn254/g1.go:1039-1118), the raw wire format -> validated limbs
    (marshal.go:826, Decoder checks :69-350) and an SRS dump streamed into HBM (kzg/marshal.go:98-113)."""
    g = gm.G1Jac("bn254")
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng([0x6E6578, 1])
    # ---- N3: out[i] = a_i * G on the device, 2^20 and 2^24 scalars
    keep = None
    for logn in (20, 24):
        n = 1 << logn
        a = uniform_scalars(rng, g, n)
        d_a = torch.from_numpy(a.view(np.int64)).cuda()
        d_p = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
        ms, _ = loop_ms(lambda: g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_p.data_ptr(), stream), 3)  # returns when complete
        c = 8 if logn < 21 else 11  # Group::batch_scalar_mul: table width by batch size
        nwin = (g.curve.fr_bits + c - 1) // c
        prods = n * (nwin * 10 + 9)  # nwin mixed additions (8M + 2S) + the shared-inversion normalisation (~9 products a point)
        # closed form: sum_i a_i G = (sum a_i) G, checked through a MultiExp with all-ones scalars against the oracle
        out[f"batch_scalar_mul_device_2p{logn}"] = {
            "ms": ms, "points_per_s": n / (ms * 1e-3), "window_bits": c, "mixed_adds_per_point": nwin,
            "mulmod_per_s": prods / (ms * 1e-3), "frac_of_measured_multiplier_rate": prods / (ms * 1e-3) / MEASURED_MULMOD_PER_S,
            "includes": "host build of the 2^(c-1) x nwin table + its upload, k_fixed_base, batch normalisation"}
        if logn == 24:
            keep = d_p
        del d_a
    # ---- N4: raw decode + checks at 2^22 (wire bytes built from the device-made points above)
    n = 1 << 22
    pts = keep[:n].cpu().numpy().view(np.uint64)  # Montgomery limbs, X | Y
    reg = np.zeros_like(pts).reshape(-1, g.curve.fp_limbs)
    rc = lib.gmsm_debug_field_op(g.gid, 0, 6, pts.ctypes.data, None, reg.shape[0], reg.ctypes.data)  # fromMont
    assert rc == 0, gm._lib.last_error()
    raw = np.ascontiguousarray(reg[:, ::-1]).byteswap().view(np.uint8).reshape(-1)  # big-endian, most significant limb first
    d_out = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    bad = _ct.c_int64(-1)
    raw_bytes = raw.size
    for level, name in ((0, "decode_only"), (2, "decode_curve_subgroup")):
        def decode(level=level):
            assert lib.gmsm_points_from_raw(g.gid, raw.ctypes.data, n, level, None, d_out.data_ptr(), _ct.byref(bad)) == 0, gm._lib.last_error()
        ms = median_ms(decode, reps=3, warm=0)
        out[f"points_from_raw_2p22_{name}"] = {"ms": ms, "GB_per_s_in": raw_bytes / (ms * 1e-3) / 1e9, "points_per_s": n / (ms * 1e-3),
                                              "source": "pageable host memory (PCIe-inclusive)"}
    same = bool((d_out.cpu().numpy().view(np.uint64) == pts).all())
    out["points_from_raw_2p22_decode_curve_subgroup"]["equal_to_source_points"] = same
    # ---- N4: the Encoder's DEFAULT (compressed) format, 2^22 points: Y = sqrt(X^3 + 3) on the device - one exponentiation of 251
    # squarings + 58 products on the lazy limbs per point (gmsm_decompress.h); the bytes are the device's own Bytes() of the points
    comp = np.zeros(n * 4 * g.aff_limbs, dtype=np.uint8)
    assert lib.gmsm_points_compress(g.gid, None, keep.data_ptr(), n, comp.ctypes.data) == 0, gm._lib.last_error()

    def decompress():
        assert lib.gmsm_points_from_compressed(g.gid, comp.ctypes.data, n, 2, None, d_out.data_ptr(), _ct.byref(bad)) == 0, gm._lib.last_error()
    ms = median_ms(decompress, reps=3, warm=0)
    prods = n * (sqrt_chain_products(g.curve.p) + 8)  # + x^3, the check y^2 = rhs, the domain changes
    peak, _ = product_peak(g)
    out["points_from_compressed_2p22"] = {
        "ms": ms, "points_per_s": n / (ms * 1e-3), "GB_per_s_in": comp.size / (ms * 1e-3) / 1e9, "products_per_point": prods // n,
        "mulmod_per_s": prods / (ms * 1e-3), "frac_of_measured_multiplier_rate": (prods / (ms * 1e-3)) / peak if peak else None,
        "equal_to_source_points": bool((d_out.cpu().numpy().view(np.uint64) == pts).all()),
        "source": "pageable host memory (PCIe-inclusive); what Decoder.Decode does per point in unsafeComputeY (marshal.go:951-989)"}
    del comp

    def validate_resident():
        assert lib.gmsm_points_validate(g.gid, None, keep.data_ptr(), n, 2, _ct.byref(bad)) == 0, gm._lib.last_error()
    ms = median_ms(validate_resident, reps=3, warm=0)
    # BN254 G1 has cofactor 1: level 2 is the curve equation alone (y^2 = x^3 + 3: three products a point) - a streaming kernel
    out["points_validate_2p22_level2_resident"] = {
        "ms": ms, "points_per_s": n / (ms * 1e-3), "GB_per_s": n * 64 / (ms * 1e-3) / 1e9, "frac_of_hbm": n * 64 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "note": "BN254 G1 is of prime order: the subgroup check is the curve equation (3 products a point on saturated limbs); "
                "the call includes its launch, the first-offender read-back and the wait (~0.05 ms)"}
    # ... and a group with a cofactor: BLS12-381 G1, [r]P = infinity by double-and-add (255 doublings + ~127 additions a point)
    g2 = gm.G1Jac("bls12_381")
    m = 1 << 20
    a2 = uniform_scalars(rng, g2, m)
    d_a2 = torch.from_numpy(a2.view(np.int64)).cuda()
    d_p2 = torch.empty((m, g2.aff_limbs), dtype=torch.int64, device="cuda")
    g2.batch_scalar_mul_device(g2.generator, d_a2.data_ptr(), m, d_p2.data_ptr(), stream)
    def validate_cofactor_group():
        assert lib.gmsm_points_validate(g2.gid, None, d_p2.data_ptr(), m, 2, _ct.byref(bad)) == 0, gm._lib.last_error()
    ms = median_ms(validate_cofactor_group, reps=3, warm=0)
    # level 2 = the reference's identity [x]([x] phi(P)) + P = 0 (gmsm_subgroup.h; x of 64 bits, weight 6): 2 x 63 doublings
    # (dbl-2008-s-1: 9 products), 5 + 1 mixed additions (madd-2008-s: 10), 5 full additions (add-2008-s: 14), phi 1, the curve
    # equation 3 - against 255 doublings + 127 additions of the definition [r]P = infinity (level 3)
    prods = m * (126 * 9 + 6 * 10 + 5 * 14 + 1 + 3)
    peak, _ = product_peak(g2)
    out["points_validate_bls12_381_g1_2p20_level2_resident"] = {
        "ms": ms, "points_per_s": m / (ms * 1e-3), "mulmod_per_s": prods / (ms * 1e-3), "products_per_point": prods // m,
        "frac_of_measured_multiplier_rate": (prods / (ms * 1e-3)) / peak if peak else None,
        "note": "on the curve and in the r-torsion by the reference's endomorphism identity (ecc/bls12-381/g1.go:481-492) on the "
                "pipeline's lazy limbs, one lane per point: compute-bound"}
    del d_a2, d_p2
    # ... and the Fp2 groups, whose group operations these kernels inline since the end of round 6: the fixed-base batch of BLS12-381 G2
    # (2^20 scalars x the G2 generator, resident) and the subgroup identity of BN254 G2 (g2.go:483-497) over its 2^20 results
    for cv, key_b, key_v in (("bls12_381", "batch_scalar_mul_bls12_381_g2_2p20", None), ("bn254", None, "points_validate_bn254_g2_2p20_level2")):
        g3 = gm.G2Jac(cv)
        a3 = uniform_scalars(rng, g3, m)
        d_a3 = torch.from_numpy(a3.view(np.int64)).cuda()
        d_p3 = torch.empty((m, g3.aff_limbs), dtype=torch.int64, device="cuda")
        ms, _ = loop_ms(lambda: g3.batch_scalar_mul_device(g3.generator, d_a3.data_ptr(), m, d_p3.data_ptr(), stream), 2)
        if key_b:
            out[key_b] = {"ms": ms, "points_per_s": m / (ms * 1e-3), "window_bits": 8, "mixed_adds_per_point": (g3.curve.fr_bits + 7) // 8}
        if key_v:
            def validate_g2():
                assert lib.gmsm_points_validate(g3.gid, None, d_p3.data_ptr(), m, 2, _ct.byref(bad)) == 0, gm._lib.last_error()
            ms = median_ms(validate_g2, reps=3, warm=0)
            out[key_v] = {"ms": ms, "points_per_s": m / (ms * 1e-3)}
        del d_a3, d_p3
    del d_out, raw, reg
    # ---- N4: SRS dump (marker | length | raw []G1Affine memory) of 2^24 points, from the page cache into HBM
    import tempfile
    n = 1 << 24
    host = keep.cpu().numpy().view(np.uint64)
    del keep
    torch.cuda.empty_cache()
    with tempfile.NamedTemporaryFile(prefix="gmsm_srs_", suffix=".dump", delete=False) as f:
        f.write(np.array([0xDEADBEEF, n], dtype=np.uint64).tobytes())
        f.write(memoryview(host).cast("B"))
        path = f.name
    try:
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rb, err = g.register_bases_dump(path, 0, True, 0, 0)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert err is None, err
            if len(ts) < 3:
                rb.release()
        ms = sorted(ts)[1]
        ones = np.zeros((4096, g.fr_limbs), dtype=np.uint64)
        ones[:] = np.array([(g.curve.fr_R % g.curve.r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(g.fr_limbs)], dtype=np.uint64)
        j_dump, _ = rb.MultiExp(ones)
        j_host, _ = g.MultiExp(host[:4096], ones)
        rb.release()
        out["bases_register_dump_2p24"] = {"ms": ms, "GB_per_s": (16 + n * 64) / (ms * 1e-3) / 1e9, "bytes": 16 + n * 64,
                                          "source": "file in the page cache -> two pinned 32 MiB buffers -> HBM -> lazy-domain rewrite",
                                          "prefix_multiexp_equal_to_host_points": bool((g.jac_to_affine(j_dump) == g.jac_to_affine(j_host)).all())}
    finally:
        os.unlink(path)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the PCIe-inclusive (host buffer) measurements")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two-in-flight (submit/collect) measurement")
    ap.add_argument("--no-also", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--shard", default="auto", choices=["auto", "windows", "points"],
                    help="N>1: decomposition of the one MultiExp over the ranks (gnark-crypto_amd/sharding.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (RCCL process group, device-side all-gather) even with one rank")
    ap.add_argument("--resident", action="store_true",
                    help="register the bases once (gmsm_bases_register) and time MultiExp over the resident form")
    ap.add_argument("--batch", type=int, default=0,
                    help="also time K MultiExp over the same registered bases spread over the ranks (replica mode)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="N>1 rehearsal on fewer devices than ranks: rank r on device r %% device_count, gloo process group")
    ap.add_argument("--also-logn", type=int, default=24, help="N>1: the sharded row that also runs through the C ABI (the 2^24 half of the metric)")
    ap.add_argument("--sharded-logns", default="22,24,26", help="N>1: sizes of the sharded rows beside the headline (north_star: 2^20-2^26)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the N3 / N4-ingest rows")
    ap.add_argument("--full-out", default=None, help="where the full record goes (default bench_full_n<N>.json beside this file, and gpurun_out/)")
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

    share = args.oversubscribe or os.environ.get("GMSM_BENCH_SHARE_DEVICE") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, RCCL over xGMI); rank 0 of the
        # relaunched job prints the JSON line on this process's stdout
        have = torch.cuda.device_count()
        if have < args.gpus and not (share and have >= 1):
            raise SystemExit(f"bench.py --gpus {args.gpus}: needs {args.gpus} devices, this machine exposes {have} "
                             "(--oversubscribe rehearses the N > 1 path on fewer)")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.dup2(json_out.fileno(), 1)  # give the real stdout back to the ranks
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (ndev <= local_rank and not share):
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, this machine exposes {ndev}")
    dev_index = local_rank % ndev if share else local_rank
    # RCCL refuses two ranks on one device: a rehearsal with more ranks than devices runs its collectives over gloo
    backend = "gloo" if (share and world > ndev) else "nccl"
    torch.cuda.set_device(dev_index)
    rank_devices = [dev_index]
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, (dev_index, ndev))
        rank_devices = [d for d, _ in seen]

    gm = importlib.import_module("gnark-crypto_amd")
    lib = gm._lib.load()
    assert lib.gmsm_set_device(dev_index) == 0, gm._lib.last_error()
    g = (gm.G1Jac if args.group == "g1" else gm.G2Jac)(args.curve)
    n = 1 << args.logn

    # synthetic, deterministic, on-curve inputs: P_i = [k0 + i*k1] G; scalars uniform
    rng = np.random.default_rng([0x6D736D, args.logn])
    k0, k1 = int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62))
    pts = g.generate_points(n, k0, k1)
    sc = uniform_scalars(rng, g, n)

    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    plan_info = g.default_plan(n)  # what the single-GPU call runs as: width, windows, GLV half scalars (2 entries per point)
    c, nwin, epp = plan_info["window_bits"], plan_info["windows"], plan_info["entries_per_point"]

    sharding = importlib.import_module("gnark-crypto_amd.sharding")
    sharded = world > 1 or args.force_dist
    plan = exchange = None
    if sharded:
        # One MultiExp over all ranks (strong scaling): point or window decomposition (sharding.py), the totals stay on
        # the device until ONE RCCL all-gather; every rank folds.
        plan = sharding.shard_plan(g, n, rank, world, args.shard)
        c, nwin, epp = plan["c"], plan["nwin"], 1  # the sharded pieces run full scalars (gmsm_window_sums_enqueue)
        exchange = sharding.Exchange(dist, torch.device("cuda", dev_index), plan["rows"], g.xyzz_limbs)
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
    prof = StageProfile(lib, level=2)
    prof.start()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jac = step()
    barrier()
    dt = time.perf_counter() - t0
    acc_only, acc_launches = prof.stop()
    prof = StageProfile(lib)  # the stage breakdown: the same steps once more, outside the timed region
    prof.start()
    for _ in range(args.steps):
        step()
    barrier()
    stages, _ = prof.stop()
    stages["accumulate"] = acc_only["accumulate"]  # the roofline's kernel duration is the timed region's
    if world > 1:
        dt = dist_max(torch, dist, dt)

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

    # Replica mode (--batch K): K scalar vectors over the same registered bases, vector j on rank j % world, each rank
    # runs its share through gmsm_multiexp_bases_batch, ONE all-gather of the K Jacobian results.
    replica = None
    if args.batch > 0:
        K = args.batch
        rbk = g.register_bases(d_points=d_pts.data_ptr(), n=n)
        d_vecs = d_sc.unsqueeze(0).repeat(len(sharding.owned_vectors(K, rank, world)) or 1, 1, 1).contiguous()

        def local_batch(mine):
            res, err = rbk.MultiExpBatch(d_scalars=d_vecs.data_ptr(), n=n, k=len(mine), stream=stream)
            assert err is None, err
            return res
        gather = (sharding.torch_all_gather(dist, torch.device("cpu") if backend == "gloo" else torch.device("cuda", dev_index))
                  if dist.is_initialized() else (lambda buf: buf[None]))
        sharding.replicated_batch(K, rank, world, g.jac_limbs, local_batch, gather)
        barrier()
        tr0 = time.perf_counter()
        resk = sharding.replicated_batch(K, rank, world, g.jac_limbs, local_batch, gather)
        barrier()
        dtr = time.perf_counter() - tr0
        if world > 1:
            dtr = dist_max(torch, dist, dtr)
        replica = {"k": K, "value": K / dtr, "unit": "MSM/s", "ms_per_msm": dtr / K * 1e3, "n_gpus": world,
                   "scaling": "weak over K (every vector is a whole MultiExp on one GPU; no data-path collective)",
                   "equal_to_serial_result": bool(all((g.jac_to_affine(r_) == g.jac_to_affine(jac)).all() for r_ in resk))}
        rbk.release()
        del d_vecs

    # PCIe-inclusive rates of the drop-in entries (SURVEY.md §8(d) "cold" / "warm-bases"): host buffers in, result out.
    # First-class fields beside `value`, never instead of it.
    host_entry = None
    if not sharded and rank == 0 and not args.no_host_entry:
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

    # The same MultiExp over registered bases WITH window tables (gmsm_bases_precompute: 2^(c w) P_i in HBM, one bucket
    # set): the resident-SRS path of kzg.Commit. Reported beside `value`, never instead of it - `value` is the entry that
    # takes the bases anew on every call, like the reference's MultiExp.
    tables = None
    if not sharded and rank == 0 and not args.no_host_entry:
        tables = tables_record(gm, g, d_pts, d_sc, sc, n, stream, jac, args.steps)

    # The same MultiExp through the drop-in C entry with the library spreading it over the devices (one process): all
    # `world` devices driven by rank 0 while the other ranks wait on the host; on one GPU two logical ranks on device 0
    # (a functional check of the path, not a speed-up: both ranks share one device and one PCIe link).
    c_abi = None
    if not args.no_host_entry:
        def through_the_c_abi():
            return c_abi_sharded(gm, g, pts, sc, rank_devices if world > 1 else [dev_index, dev_index], g.jac_to_affine(jac))
        if world > 1:
            barrier()
            c_abi = host_side_wait(dist, rank, "c_abi_headline", through_the_c_abi)
        elif not sharded:
            c_abi = through_the_c_abi()

    # N > 1: the 2^24 half of BASELINE.json's metric, sharded the same way (every rank takes part; rank 0 reports)
    also_sharded, sharded_rows = None, []
    headline_parts = None
    if sharded:
        # the headline's stage times (max over ranks) and its exchange, cut at the joints (outside the timed loop)
        stages = max_over_ranks(dist, world, stages)
        headline_parts = max_over_ranks(dist, world, exchange_breakdown(torch, g, sharding, plan, enqueue, exchange))
    if sharded and not args.no_also and (args.curve, args.group) == ("bn254", "g1"):
        del d_pts_loc, d_sc_loc
        logns = sorted({int(x) for x in args.sharded_logns.split(",") if x} | {args.also_logn})
        for ln in logns:
            row = sharded_also(gm, lib, torch, dist, sharding, rank, world, dev_index, args.shard, rank_devices, logn=ln,
                               steps=3 if ln >= 26 else 5, with_c_abi=(ln == args.also_logn))
            sharded_rows.append(row)
            if ln == args.also_logn:
                also_sharded = row

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.steps / dt
        # this rank's share of the (point, window) pairs: all windows of its slice, or its windows of all points
        my_pairs = (plan["hi"] - plan["lo"]) * len(range(plan["win_first"], nwin, plan["win_stride"])) if sharded else n * nwin * epp
        out = {
            "metric": "G1 MSM/sec (BN254)" if (args.curve, args.group) == ("bn254", "g1") else f"{args.group.upper()} MSM/sec ({args.curve})",
            "value": value, "unit": "MSM/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.curve.upper()} {args.group.upper()} MultiExp 2^{args.logn} points, bases+scalars resident in HBM",
                       "arithmetic": f"{g.curve.p.bit_length()}-bit Montgomery field on 32-bit words (lazy 28/29-bit limbs, v_mad_u64_u32)",
                       "points": n, "window_bits": c, "windows": nwin, "glv_half_scalars": epp == 2, "resident_bases": resident is not None,
                       "parallelism": "single GPU" if not sharded else
                                      f"{plan['mode']}-sharded x{world} + one {'RCCL' if backend == 'nccl' else backend} all-gather"},
            "value_cold": host_entry["cold_msm_per_s"] if host_entry else None,
            "value_warm_bases": host_entry["warm_bases_msm_per_s"] if host_entry else None,
            "points_per_s": value * n,
            "stage_ms": stages,
            "stage_ms_note": STAGE_NOTE,
            "pipelined": pipelined,
            "host_entry": host_entry,
            "value_tables": tables["device_scalars_msm_per_s"] if tables else None,
            "tables": tables,
            "c_abi_sharded": c_abi,
            "replica_batch": replica,
            "roofline": roofline_record(g, n, my_pairs / (n * nwin * epp), stages, acc_launches,
                                        measured_traffic(args.curve, args.group, args.logn, world, nwin)),
            "int_roofline": int_roofline_record(g, my_pairs, stages["accumulate"]),
        }
        tail = {}
        if sharded:
            # what the process group really was: the SCALE record shows that RCCL saw N ranks on N devices
            out["backend"] = dist.get_backend()
            out["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
            out["devices_seen"] = rank_devices
            out["device_count"] = ndev
            out["oversubscribed"] = bool(world > ndev)
            # the sharded result against the same MultiExp computed by this rank alone (after the timed region)
            single = g.multiexp_device(d_pts.data_ptr(), d_sc.data_ptr(), n, stream)
            out["equal_to_single_gpu_result"] = bool((g.jac_to_affine(single) == g.jac_to_affine(jac)).all())
            if headline_parts is not None:
                out["compute_ms"] = round(headline_parts["compute_ms"], 4)
                out["exchange_ms"] = round(headline_parts["gather_ms"] + headline_parts["fold_ms"], 4)
                out["exchange_parts_ms"] = {"all_gather_and_d2h": round(headline_parts["gather_ms"], 4), "fold": round(headline_parts["fold_ms"], 4)}
                out["stage_ms_note"] = "max over the ranks; " + STAGE_NOTE
            if also_sharded is not None:
                out["also"] = [r for r in sharded_rows if r is not also_sharded] + [also_sharded]  # the 2^24 row last
                cab = also_sharded.get("c_abi_sharded") or {}
                out["n24"] = {"workload": also_sharded["workload"], "ms_per_step": round(also_sharded["ms_per_step"], 4),
                              "value": round(also_sharded["value"], 3), "bit_exact": also_sharded.get("bit_exact"),
                              "compute_ms": also_sharded["compute_ms"], "exchange_ms": also_sharded["exchange_ms"],
                              "sharded_rows": {r["workload"].split(" points")[0].split()[-1]: [round(r["ms_per_step"], 3), r.get("bit_exact")]
                                               for r in sharded_rows},  # "2^22": [ms_per_step, bit_exact], ...
                              "c_abi_cold_ms": cab.get("cold_ms"), "c_abi_warm_bases_ms": cab.get("warm_bases_ms"),
                              "c_abi_equal_to_reference_result": cab.get("equal_to_reference_result")}
            tail = {"backend": out["backend"], "rccl_ranks": out["rccl_ranks"], "devices_seen": rank_devices,
                    "equal_to_single_gpu_result": out["equal_to_single_gpu_result"],
                    "c_abi_equal_to_reference_result": (c_abi or {}).get("equal_to_reference_result")}
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline(g, pts, sc, jac, args.curve, args.group))
        if world == 1 and not sharded and not args.no_host_entry:
            out["first_call"] = first_call_record(args.curve, args.group)
        if world == 1 and not sharded and not args.no_also:
            del d_pts, d_sc, pts
            torch.cuda.empty_cache()
            out["fft"] = [fft_config(gm, torch, "bn254", 20), fft_config(gm, torch, "bn254", 24),
                          fft_config(gm, torch, "bls12_381", 24, reps=3), fft_config(gm, torch, "bw6_761", 24, reps=3),
                          fft_extra(gm, torch, "bn254", 24)]
            if not args.no_next_rows:
                out["next_rows"] = next_rows(gm, lib, torch)
                torch.cuda.empty_cache()
            # the reference's own benchmark matrix: every size under skewed scalar distributions, and the sizes below 2^20
            out["distributions"] = distributions_block(gm, lib, torch)
            g2 = distributions_block(gm, lib, torch, configs=(("bls12_381", "g2", 22, 3),), kinds=["uniform", "smallvalues"], cold=False)
            out["distributions"]["rows"] += g2["rows"]
            out["distributions"]["worst_vs_uniform"] = max(out["distributions"]["worst_vs_uniform"], g2["worst_vs_uniform"])
            out["distributions"]["all_bit_exact"] = out["distributions"]["all_bit_exact"] and g2["all_bit_exact"]
            out["small_n"] = small_n_block(gm, torch)
            # ... and 2^5 points (the reference's smallest benchmark size) for the other groups: resident ms, checked against the port
            wide = {}
            for cv, gp in (("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2"), ("bw6_761", "g1")):
                blk = small_n_block(gm, torch, cv, gp, logns=(5,), reps=20, with_cpu=False)
                wide[f"{cv}_{gp}"] = blk["rows"][0]["resident_ms"] if blk["rows"][0]["bit_exact"] else None
            out["small_n"]["wide"] = wide
            out["also"] = [also_config(gm, lib, torch, *cfg_) for cfg_ in ALSO
                           if (cfg_[0], cfg_[1], cfg_[2]) != (args.curve, args.group, args.logn)]
            r24 = next((r for r in out["also"] if r["workload"].startswith("BN254 G1 MultiExp 2^24")), None)
            if r24 is not None:  # the 2^24 half of BASELINE.json's metric once more, compact, where the driver's tail keeps it
                out["n24"] = {"ms_per_step": round(r24["ms_per_step"], 4), "value": round(r24["value"], 3),
                              "value_warm_bases": round(r24.get("value_warm_bases", 0.0), 3), "value_cold": round(r24.get("value_cold", 0.0), 3),
                              "roofline_frac": round(r24["roofline"]["frac"], 5), "roofline_traffic": r24["roofline"]["traffic"],
                              "accumulate_ms": round(r24["roofline"]["avg_launch_ms"], 4), "window_bits": r24["window_bits"],
                              "cpu_baseline_value": round(r24.get("cpu_baseline", {}).get("value", 0.0), 4),
                              "cpu_cores": r24.get("cpu_baseline", {}).get("cores"), "bit_exact": r24["bit_exact"],
                              "bit_exact_vs_cpu_port": r24.get("bit_exact_vs_cpu_port")}
        # the headline's PCIe-inclusive rates and its parity verdict go last as well
        for k in ("value_cold", "value_warm_bases", "bit_exact"):
            if k in out:
                tail[k] = out[k]
        if "distributions" in out:
            tail["distributions_worst_vs_uniform"] = out["distributions"]["worst_vs_uniform"]
            tail["distributions_all_bit_exact"] = out["distributions"]["all_bit_exact"]
        if "small_n" in out:
            tail["small_n_ms"] = {f"2^{r['logn']}": r["resident_ms"] for r in out["small_n"]["rows"]}
        tail.update({"value": round(value, 3), "ms_per_step": round(ms_per_step, 4), "n_gpus": world})
        out["tail"] = tail
        emit(out, world, args.full_out, json_out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def emit(full, world, full_out, json_out):
    """Write the full record to the side file(s) and print the slim line (tools/bench_line.py: at most MAX_LINE_BYTES, with the
    contract keys, roofline, int_roofline, cpu_baseline and one compact row per other configuration)."""
    from tools import bench_line
    paths = [full_out or os.path.join(ROOT, f"bench_full_n{world}.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not full_out:  # on the GPU box: what gpurun merges back
        paths.append(os.path.join(ROOT, "gpurun_out", f"bench_full_n{world}.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f)
            written = written or os.path.relpath(p, ROOT)
        except OSError as e:  # a read-only tree must not cost the line
            print(f"[bench] could not write {p}: {e}", file=sys.stderr)
    line = bench_line.encode(bench_line.fit(bench_line.slim_line(full, full_path=written)))
    assert len(line.encode()) <= bench_line.MAX_LINE_BYTES, len(line)
    for problem in bench_line.validate(line):
        print(f"[bench] line problem: {problem}", file=sys.stderr)
    print(line, file=json_out, flush=True)


def first_call_record(curve, group):
    """Cold start of the drop-in entry, measured in a FRESH process (tools/first_call.py: dlopen, HIP runtime + device, the first
    2^10 and 2^20 calls, another group's first call); None when the probe fails."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_call.py"), curve, group], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, GMSM_NO_TORCH="1"))
        line = next(ln for ln in p.stdout.splitlines() if ln.startswith("{"))
        return json.loads(line)
    except (subprocess.SubprocessError, StopIteration, ValueError, OSError) as e:
        print(f"[bench] first_call probe failed: {e}", file=sys.stderr)
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


def cpu_baseline(g, pts, sc, gpu_jac, curve="bn254", group="g1", min_seconds=10.0):
    """The oracle = C restatement of gnark-crypto's MultiExp (bestC, split recursion, one task per (leaf, window),
    extended-Jacobian / batch-affine buckets), on all host cores.  Bounded: repeats whole MSMs until ~min_seconds have
    elapsed (at least 1)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # test infrastructure, used here only as the reported baseline and the checker
    o = oracle.Oracle(curve, group)
    cores = effective_cpus()
    threads = min(2 * cores, len(os.sched_getaffinity(0)))  # 2 software threads per allowed core measured best
    reps, t_total, jac = 0, 0.0, None
    while reps < 1 or (t_total < min_seconds and reps < 50):
        t0 = time.perf_counter()
        err, jac = o.multiexp(pts, sc, nb_tasks=0, num_cpu=cores, nthreads=threads)
        t_total += time.perf_counter() - t0
        reps += 1
        assert err == 0
    exact = bool((o.jac_to_affine(jac) == g.jac_to_affine(gpu_jac)).all())
    mul_ns = oracle.Field(f"{curve}_fp", g.curve.fp_limbs).mul_ns()
    return {
        "cpu_baseline": {"value": reps / t_total, "unit": "MSM/s", "cores": cores, "kind": "port",
                         "threads": threads, "base_field_mul_ns": round(mul_ns, 1), "batch_affine": True,
                         "sample": f"{reps} full MSM(s) of the same 2^{int(np.log2(len(pts)))} input, {t_total:.1f} s total; "
                                   + getattr(oracle, "BASELINE_NOTE", "C restatement of gnark-crypto's algorithm (ext-Jacobian buckets, batch-affine off)")},
        "bit_exact": exact,
    }


import ctypes as _ct  # noqa: E402

if __name__ == "__main__":
    main()
