#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats (csv) -> one markdown table per profiled command.
usage: python tools/kernel_stats_md.py <label>=<dir-with-*_kernel_stats.csv> ... > profiles/rNN_kernel_stats.md"""
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("gmsm::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("unsigned short", "u16").replace("unsigned int", "u32")
    return name if len(name) < 100 else name[:97] + "..."


def main(args):
    print("# rocprofv3 --kernel-trace --stats summaries\n")
    print("Each block is one command, profiled on its own; durations in microseconds. `bench.py`'s `roofline.avg_launch_ms` "
          "(HIP events inside libgmsm) is to be compared with the `k_accumulate_seg` row of the matching block.\n")
    for a in args:
        label, d = a.split("=", 1)
        files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if not files:
            print(f"## {label}\n\n(no kernel_stats.csv under {d})\n")
            continue
        rows = list(csv.DictReader(open(files[0])))
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"## {label}\n")
        print(f"kernel time total {total / 1e6:.3f} ms\n")
        print("| kernel | calls | total ms | avg us | min us | max us | % |")
        print("|---|---|---|---|---|---|---|")
        for r in rows:
            if float(r["TotalDurationNs"]) / total < 0.0005:
                continue
            print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                  f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
