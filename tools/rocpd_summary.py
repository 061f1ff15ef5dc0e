#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) kernel trace into a per-kernel table (calls, total/avg/min/max duration).
Usage: python tools/rocpd_summary.py gpurun_out/prof_xxx/bench_results.db > profiles/xxx_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    stats = {}
    for name, s, e in rows:
        d = (e - s) / 1e3  # ns -> us
        st = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    print(f"source: {path}")
    print(f"kernel dispatches: {len(rows)}, total kernel time {total / 1e3:.3f} ms\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, (c, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {c} | {t / 1e3:.3f} | {t / c:.1f} | {mn:.1f} | {mx:.1f} | {100 * t / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
