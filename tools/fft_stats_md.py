#!/usr/bin/env python3
"""gpurun_out/<dir> of tools/profile_fft.sh -> markdown: per-kernel time table (rocprofv3 --kernel-trace --stats) and the
PMC counters of the k_fft_pass_lz launches, grouped by launch geometry (grid size = which pass of which transform size).
usage: python tools/fft_stats_md.py gpurun_out/fft_prof > profiles/r04_fft_stats.md"""
import collections
import csv
import glob
import sys


def stats_table(d):
    rows = []
    for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:12]:
        name = r["Name"].split("(")[0][:90]
        out.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                   f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    return "\n".join(out)


def trace_by_grid(d):
    """avg duration of k_fft_pass_lz launches keyed by (grid, workgroup, lds)"""
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fft_pass_lz" not in r["Kernel_Name"]:
                continue
            key = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), int(r.get("LDS_Block_Size", 0) or 0))
            per[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def counters(d, name):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fft_pass_lz" not in r["Kernel_Name"]:
                continue
            key = (int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])), int(r["Workgroup_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
            per[key][(r["Counter_Name"], int(r["Dispatch_Id"]))].append(float(r["Counter_Value"]))
    out = {}
    for key, m in per.items():
        agg = collections.defaultdict(list)
        for (cn, _), vals in m.items():
            agg[cn].append(sum(vals))
        out[key] = {cn: sum(v) / len(v) for cn, v in agg.items()}
    return out


def main(d):
    print("## rocprofv3 --kernel-trace --stats (all launches of the command)\n")
    print(stats_table(d))
    tr = trace_by_grid(d)
    if tr:
        print("\n## k_fft_pass_lz by launch geometry (trace)\n")
        print("| workgroups | threads | LDS B | launches | avg us | min us |")
        print("|---|---|---|---|---|---|")
        for key in sorted(tr):
            v = tr[key]
            print(f"| {key[0]} | {key[1]} | {key[2]} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} |")
    for name in ("sq_time", "sq_insts", "lds", "fetch", "write"):
        c = counters(d, name)
        if not c:
            continue
        names = sorted({cn for m in c.values() for cn in m})
        print(f"\n## PMC pass `{name}` (mean per launch, summed over the chip)\n")
        print("| workgroups | threads | " + " | ".join(names) + " |")
        print("|---|---|" + "---|" * len(names))
        for key in sorted(c):
            print(f"| {key[0]} | {key[1]} | " + " | ".join(f"{c[key].get(n, float('nan')):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
