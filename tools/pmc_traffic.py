#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of the same single-configuration bench.py command)
-> profiles/traffic_rNN.json: HBM/fabric bytes per k_accumulate_seg launch, keyed curve_group_logn.

usage: python tools/pmc_traffic.py <key>=<dir> ... > profiles/traffic_r02.json
       each <dir> holds fetch/*counter_collection.csv and write/*counter_collection.csv of one configuration

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE under-counts wide coalesced streaming reads by 2x
(MI355X_MICROARCH.md, HBM section); this kernel reads 16-byte pieces of scattered 64..192-byte records and 4-byte
sequential entries, a pattern the guide calls uncalibrated: the value is reported as measured."""
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
    out, detail = {}, {}
    for a in args:
        key, d = a.split("=", 1)
        f = glob.glob(d + "/fetch/*counter_collection.csv")
        w = glob.glob(d + "/write/*counter_collection.csv")
        if not f or not w:
            continue
        fb, nf = per_launch_mean(f[0], "FETCH_SIZE")
        wb, nw = per_launch_mean(w[0], "WRITE_SIZE")
        if fb is None or wb is None:
            continue
        out[key] = fb + wb
        detail[key] = {"fetch_bytes": fb, "write_bytes": wb, "launches": [nf, nw]}
    print(json.dumps({"note": "HBM/fabric bytes per k_accumulate_seg launch: rocprofv3 PMC FETCH_SIZE + WRITE_SIZE (separate passes, "
                              "KiB x 1024) on the round-2 build, one bench.py configuration per pass; FETCH_SIZE as reported (gfx950 may "
                              "under-count streaming reads by up to 2x, MI355X_MICROARCH.md)",
                      "k_accumulate_seg": out, "detail": detail}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
