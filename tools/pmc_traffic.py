#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of the same single-configuration bench.py command)
-> profiles/traffic_rNN.json: HBM/fabric bytes per k_accumulate_seg launch, keyed curve_group_logn.

usage: python tools/pmc_traffic.py <key>=<dir> ... > profiles/traffic_r02.json
       each <dir> holds fetch/*counter_collection.csv and write/*counter_collection.csv of one configuration

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE under-counts wide coalesced streaming reads by 2x
(MI355X_MICROARCH.md, HBM section).  This kernel's traffic is the gather of one 64..192-byte record per lane (plus 4-byte
sequential entries, ~6 %): calibrated in round 3 with tools/ubench_gather.hip on exactly that pattern - 2^24 lanes x one
64-byte record out of a 4 GiB table report FETCH_SIZE = 1024.0 MiB for 1024.0 MiB read (factor 1.00), the 16-byte-per-lane
stream over the same table 2048 MiB for 4096 MiB (the guide's factor 2) - profiles/r03_fetch_calibration.md.  The values
are therefore reported as measured."""
import collections
import csv
import glob
import json
import sys


def per_launch_mean(path, counter):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "k_accumulate_seg" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = list(per.values())
    return (sum(vals) / len(vals) * 1024, len(vals)) if vals else (None, 0)


def main(args):
    out, detail, windows = {}, {}, {}
    for a in args:
        key, d = a.split("=", 1)
        f = glob.glob(d + "/fetch/**/*counter_collection.csv", recursive=True)
        w = glob.glob(d + "/write/**/*counter_collection.csv", recursive=True)
        if not f or not w:
            continue
        fb, nf = per_launch_mean(f[0], "FETCH_SIZE")
        wb, nw = per_launch_mean(w[0], "WRITE_SIZE")
        if fb is None or wb is None:
            continue
        out[key] = fb + wb
        detail[key] = {"fetch_bytes": fb, "write_bytes": wb, "launches": [nf, nw]}
        try:  # window count of the profiled run: bench.py reports the figure only for a run with the same geometry
            windows[key] = json.load(open(d + "/bench.json"))["config"]["windows"]
        except (OSError, ValueError, KeyError):
            pass
    print(json.dumps({"note": "HBM/fabric bytes per k_accumulate_seg launch: rocprofv3 PMC FETCH_SIZE + WRITE_SIZE (separate passes, "
                              "KiB x 1024) on this round's build with the shipped window table, one bench.py configuration per "
                              "pass; FETCH_SIZE as reported - calibrated at factor 1.00 for this kernel's 64-byte-per-lane gather "
                              "(profiles/r03_fetch_calibration.md)",
                      "k_accumulate_seg": out, "windows": windows, "detail": detail}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
