"""The ONE JSON line bench.py prints, assembled from the full record of a run.

The driver parses the line it reads from stdout and keeps a bounded copy: round 5's N = 1 line had grown to 26 KB and came back
`parsed: null`. bench.py therefore builds the full record as before, writes it to a side file (bench_full_n<N>.json, copied under
profiles/ per round), and prints slim_line(full): the contract keys, `roofline`, `int_roofline`, `cpu_baseline`, the headline's
stage times and one COMPACT row per other configuration. MAX_LINE_BYTES is enforced here, by bench.py before it prints, and by
tests/test_bench_line.py on a canned record.

    python tools/bench_line.py <line.json | ->         validate a printed line (schema + size); exit code 1 on a violation
    python tools/bench_line.py --slim <full.json>      print the slim line of a stored full record
"""
import json
import sys

MAX_LINE_BYTES = 6144

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample")

STAGE_NOTE = "accumulate = HIP events around k_accumulate_seg inside the timed region; other stages: same steps re-run outside it"


def _r(x, digits=4):
    """Round floats (significant figures for small values, so that 0.0107 does not become 0.01); leave the rest."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        if x == 0.0:
            return 0.0
        if abs(x) >= 1e6:
            return float(f"{x:.5g}")
        if abs(x) >= 100:
            return round(x, 2)
        if abs(x) >= 1:
            return round(x, digits)
        return float(f"{x:.4g}")
    return x


def _rd(d, keys=None):
    return {k: _r(v) for k, v in d.items() if (keys is None or k in keys) and not isinstance(v, (dict, list))}


def _short_workload(w):
    """'BN254 G1 MultiExp 2^22 points, bases+scalars resident in HBM' -> 'bn254_g1_2^22'."""
    parts = w.replace(",", "").split()
    try:
        size = next(p for p in parts if p.startswith("2^"))
        return f"{parts[0].lower()}_{parts[1].lower()}_{size}"
    except StopIteration:
        return w[:40]


def compact_roofline(rf):
    if not rf:
        return None
    out = _rd(rf, ROOFLINE_KEYS + ("kernel", "avg_launch_ms", "launches_per_msm"))
    if "algorithmic_bytes_per_launch" in rf:
        out["bytes_per_launch"] = int(rf["algorithmic_bytes_per_launch"])
    if rf.get("traffic") is not None:
        out["traffic"] = int(rf["traffic"])
    return out


def compact_also_row(r):
    """One other BASELINE configuration: workload, ms, roofline.frac / traffic, the integer fraction, the CPU port, parity."""
    rf = r.get("roofline") or {}
    ir = r.get("int_roofline") or {}
    cb = r.get("cpu_baseline") or {}
    row = {"workload": _short_workload(r.get("workload", "")), "ms": _r(r.get("ms_per_step")), "c": r.get("window_bits"),
           "acc_ms": _r(rf.get("avg_launch_ms")), "launches": rf.get("launches_per_msm"),
           "roofline": {"frac": _r(rf.get("frac")), "achieved": _r(rf.get("achieved")),
                        "traffic": int(rf["traffic"]) if rf.get("traffic") is not None else None},
           "int_frac": _r(ir.get("frac_of_measured")), "cpu_value": _r(cb.get("value")), "bit_exact": r.get("bit_exact")}
    if "value_cold" in r:
        row["cold_ms"] = _r(r.get("cold_ms"))
        row["warm_bases_ms"] = _r(r.get("warm_bases_ms"))
    if "n_gpus" in r:  # a sharded row of the N > 1 line
        row.update({"n_gpus": r["n_gpus"], "compute_ms": _r(r.get("compute_ms")), "exchange_ms": _r(r.get("exchange_ms"))})
        if r.get("c_abi_sharded"):
            cab = r["c_abi_sharded"]
            row["c_abi_sharded"] = {"cold_ms": _r(cab.get("cold_ms")), "warm_bases_ms": _r(cab.get("warm_bases_ms")),
                                    "equal": cab.get("equal_to_reference_result")}
        for k in ("roofline", "int_frac", "cpu_value", "acc_ms", "launches"):
            if row.get(k) is None or (k == "roofline" and row[k]["frac"] is None):
                row.pop(k, None)
        if r.get("stage_ms"):
            row["stage_ms"] = {k: _r(v) for k, v in r["stage_ms"].items()}
    return row


def slim_line(full, full_path=None):
    """The dict bench.py prints: every key the driver's contract names, the three measurement blocks, compact rows."""
    out = {k: full.get(k) for k in CONTRACT_KEYS}
    out["value"] = _r(out["value"])
    out["ms_per_step"] = _r(out["ms_per_step"])
    cfg = dict(full.get("config") or {})
    cfg.pop("arithmetic", None)
    out["config"] = cfg
    out["arithmetic"] = "Montgomery field on lazy 28/29-bit limbs in u32 words (v_mad_u64_u32)"
    for k in ("value_cold", "value_warm_bases", "value_tables", "bit_exact", "backend", "rccl_ranks", "devices_seen", "device_count",
              "oversubscribed", "equal_to_single_gpu_result", "compute_ms", "exchange_ms"):
        if k in full:
            out[k] = _r(full[k])
    out["stage_ms"] = {k: _r(v) for k, v in (full.get("stage_ms") or {}).items() if k != "reserved"}
    out["stage_ms_note"] = STAGE_NOTE
    out["roofline"] = compact_roofline(full.get("roofline"))
    ir = full.get("int_roofline")
    if ir:
        out["int_roofline"] = {"unit": "field products/s", "achieved": _r(ir.get("achieved_mulmod_per_s")),
                               "peak_nominal": _r(ir.get("peak_mulmod_per_s")), "peak_measured": _r(ir.get("measured_peak_mulmod_per_s")),
                               "frac": _r(ir.get("frac")), "frac_of_measured": _r(ir.get("frac_of_measured"))}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "threads": cb.get("threads"), "sample": str(cb.get("sample", ""))[:160]}
    p = full.get("pipelined")
    if p:
        out["pipelined"] = {"value": _r(p.get("value")), "in_flight": p.get("in_flight"),
                            "two_blocking_callers": _r((p.get("two_blocking_callers") or {}).get("value")),
                            "batch_call_host_scalars": _r((p.get("batch_call_host_scalars") or {}).get("value")),
                            "equal_to_serial_result": bool(p.get("equal_to_serial_result")
                                                           and (p.get("two_blocking_callers") or {}).get("equal_to_serial_result", True)
                                                           and (p.get("batch_call_host_scalars") or {}).get("equal_to_serial_result", True))}
    t = full.get("tables")
    if t:
        out["tables"] = {"c": t.get("window_bits"), "ms": _r(t.get("device_scalars_ms")), "host_scalars_ms": _r(t.get("host_scalars_ms")),
                         "two_in_flight_ms": _r(t.get("two_in_flight_ms")), "build_ms": _r(t.get("build_ms")),
                         "equal": t.get("equal_to_headline_result")}
    c = full.get("c_abi_sharded")
    if c:
        out["c_abi_sharded"] = {"devices": c.get("devices"), "cold_ms": _r(c.get("cold_ms")), "warm_bases_ms": _r(c.get("warm_bases_ms")),
                                "equal": c.get("equal_to_reference_result")}
    if full.get("replica_batch"):
        rb = full["replica_batch"]
        out["replica_batch"] = {"k": rb.get("k"), "value": _r(rb.get("value")), "equal": rb.get("equal_to_serial_result")}
    if full.get("first_call"):
        out["first_call"] = {k: _r(v) for k, v in full["first_call"].items() if not isinstance(v, (dict, list, str))}
    if full.get("also"):
        out["also"] = [compact_also_row(r) for r in full["also"]]
    d = full.get("distributions")
    if d:
        rows = {}
        for r in d.get("rows", []):
            rows.setdefault(f"{r['group']}_2^{r['logn']}", {})[r["distribution"]] = r.get("vs_uniform")
        out["distributions"] = {"vs_uniform": rows, "worst": d.get("worst_vs_uniform"), "all_bit_exact": d.get("all_bit_exact")}
    s = full.get("small_n")
    if s:
        out["small_n"] = {"group": s.get("group"), "resident_ms": {str(r["logn"]): _r(r["resident_ms"]) for r in s["rows"]},
                          "tables_ms": {str(r["logn"]): _r(r.get("registered_tables_ms")) for r in s["rows"] if r["logn"] <= 12},
                          "crossover_logn": s.get("crossover_logn_cold_vs_cpu_port"),
                          "all_bit_exact": all(r.get("bit_exact") for r in s["rows"])}
        if s.get("wide"):
            out["small_n"]["wide_2^5_ms"] = {k: _r(v) for k, v in s["wide"].items()}
    if full.get("fft"):
        out["fft_ms"] = {}
        for r in full["fft"]:
            if "ms" in r:
                w = r["workload"].split()
                out["fft_ms"][f"{w[0].lower()}_{next(p for p in w if p.startswith('2^'))}"] = _r(r["ms"])
        out["fft_round_trips_exact"] = all(r.get("round_trip_exact", True) for r in full["fft"])
    nr = full.get("next_rows")
    if nr:
        out["next_rows_ms"] = {k: _r(v.get("ms")) for k, v in nr.items() if isinstance(v, dict) and "ms" in v}
    if full.get("n24"):
        out["n24"] = {k: _r(v) for k, v in full["n24"].items() if not isinstance(v, (dict, list)) or k == "sharded_rows"}
    if full_path:
        out["full_record"] = full_path
    if full.get("tail"):
        out["tail"] = full["tail"]
    return out


def encode(line_dict):
    return json.dumps(line_dict, separators=(",", ":"))


def fit(line_dict):
    """Drop the least important blocks until the encoded line fits MAX_LINE_BYTES (never the contract or measurement blocks)."""
    dropped = []
    for k in ("next_rows_ms", "fft_ms", "c_abi_sharded", "tables", "small_n", "distributions", "stage_ms_note", "first_call"):
        if len(encode(line_dict)) <= MAX_LINE_BYTES:
            break
        if k in line_dict:
            line_dict.pop(k)
            dropped.append(k)
    if dropped:
        line_dict["dropped_for_size"] = dropped
    return line_dict


def validate(line, n1_requirements=True):
    """Problems of a printed line (str or dict) as a list of strings; empty = fine."""
    problems = []
    if isinstance(line, str):
        if len(line.encode()) > MAX_LINE_BYTES:
            problems.append(f"line is {len(line.encode())} bytes > {MAX_LINE_BYTES}")
        if "\n" in line.strip():
            problems.append("more than one line")
        try:
            d = json.loads(line)
        except ValueError as e:
            return problems + [f"not JSON: {e}"]
    else:
        d = line
        if len(encode(d).encode()) > MAX_LINE_BYTES:
            problems.append(f"line is {len(encode(d).encode())} bytes > {MAX_LINE_BYTES}")
    for k in CONTRACT_KEYS:
        if k not in d:
            problems.append(f"missing contract key {k}")
    if not isinstance(d.get("config"), dict) or "workload" not in d.get("config", {}):
        problems.append("config.workload missing")
    if isinstance(d.get("config"), dict) and "model" in d["config"]:
        problems.append("config carries a model key")
    if not isinstance(d.get("value"), (int, float)) or not d.get("value", 0) > 0:
        problems.append("value is not a positive number")
    if d.get("value") and d.get("ms_per_step") and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) > 0.02:
        problems.append("value and ms_per_step disagree")
    if d.get("vs_baseline") is not None:
        problems.append("vs_baseline must be null (BASELINE.md holds no published number)")
    if n1_requirements and d.get("n_gpus") == 1:
        rf = d.get("roofline")
        if not isinstance(rf, dict):
            problems.append("roofline missing")
        else:
            for k in ROOFLINE_KEYS:
                if k not in rf:
                    problems.append(f"roofline.{k} missing")
            if rf.get("achieved") and rf.get("peak") and abs(rf["achieved"] / rf["peak"] - rf.get("frac", 0)) > 1e-3:
                problems.append("roofline.frac != achieved / peak")
            if rf.get("bound") not in ("hbm", "mfma"):
                problems.append("roofline.bound")
        cb = d.get("cpu_baseline")
        if not isinstance(cb, dict):
            problems.append("cpu_baseline missing")
        else:
            for k in CPU_BASELINE_KEYS:
                if k not in cb:
                    problems.append(f"cpu_baseline.{k} missing")
            if cb.get("kind") not in ("reference", "port"):
                problems.append("cpu_baseline.kind")
        if not isinstance(d.get("int_roofline"), dict):
            problems.append("int_roofline missing")
        if d.get("bit_exact") is not True:
            problems.append("bit_exact is not true")
        for r in d.get("also") or []:
            if r.get("bit_exact") is not True:
                problems.append(f"also row {r.get('workload')} is not bit_exact")
    return problems


def main(argv):
    if len(argv) >= 2 and argv[0] == "--slim":
        with open(argv[1]) as f:
            print(encode(fit(slim_line(json.load(f), full_path=argv[1]))))
        return 0
    src = sys.stdin.read() if (not argv or argv[0] == "-") else open(argv[0]).read()
    lines = [ln for ln in src.splitlines() if ln.strip().startswith("{")]
    if len(lines) != 1:
        print(f"expected ONE JSON line, found {len(lines)}")
        return 1
    problems = validate(lines[0])
    for p in problems:
        print("PROBLEM:", p)
    print(f"{len(lines[0].encode())} bytes, {'ok' if not problems else str(len(problems)) + ' problem(s)'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
