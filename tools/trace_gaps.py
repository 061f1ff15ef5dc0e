#!/usr/bin/env python3
"""rocprofv3 kernel trace (csv) of a bench.py run -> where the library's HIP events and the trace may differ for
k_accumulate_seg: the kernel's own duration, the idle gap before it (end of the previous kernel of the pipeline to its
start) and after it (its end to the start of the fix-up).  usage: python tools/trace_gaps.py <dir>"""
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur, before, after, conv_overlap = [], [], [], 0
    for i, r in enumerate(rows):
        if "k_accumulate_seg" not in r["Kernel_Name"]:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        prev = max((int(x["End_Timestamp"]) for x in rows[max(0, i - 6):i] if int(x["End_Timestamp"]) <= s), default=s)
        nxt = min((int(x["Start_Timestamp"]) for x in rows[i + 1:i + 4] if int(x["Start_Timestamp"]) >= e), default=e)
        dur.append((e - s) / 1e3)
        before.append((s - prev) / 1e3)
        after.append((nxt - e) / 1e3)
    n = len(dur)
    print(f"{n} launches of k_accumulate_seg: duration avg {sum(dur) / n:.1f} us (min {min(dur):.1f}, max {max(dur):.1f}); "
          f"idle before avg {sum(before) / n:.1f} us, after avg {sum(after) / n:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
