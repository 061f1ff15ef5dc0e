"""Side-by-side of the bench lines tools/gpu_session7.sh wrote: signed-limb mixed addition against the unsigned build."""
import glob, json, os, sys

O = sys.argv[1]


def load(p):
    with open(p) as f:
        for line in f:
            line = line.strip()
            if line.startswith("{"):
                return json.loads(line)
    return None


def rows(j):
    out = {"headline ms": j["ms_per_step"], "accumulate ms (events)": (j.get("roofline") or {}).get("avg_launch_ms"),
           "tables MSM/s": j.get("value_tables"), "pipelined MSM/s": (j.get("pipelined") or {}).get("value")}
    for a in j.get("also", []) or []:
        out[a["workload"].split(",")[0] + " ms"] = a.get("ms_per_step")
        acc = (a.get("roofline") or {}).get("avg_launch_ms")
        if acc:
            out[a["workload"].split(",")[0] + " accumulate ms"] = acc
    return out


KINDS = ("unsigned", "flushloop", "signed")
res = {}
for kind in KINDS:
    for p in sorted(glob.glob(os.path.join(O, "bench_%s_*.json" % kind))):
        j = load(p)
        if j:
            res.setdefault(kind, []).append(rows(j))
keys = []
for kind in res:
    for r in res[kind]:
        for k in r:
            if k not in keys:
                keys.append(k)
print("%-46s %10s %10s %10s %7s %7s" % ("row (mean of runs)", "unsigned", "flushloop", "signed", "fl/un", "sg/un"))
for k in keys:
    m = {}
    for kind in KINDS:
        v = [r[k] for r in res.get(kind, []) if isinstance(r.get(k), (int, float))]
        m[kind] = sum(v) / len(v) if v else None
    if m["unsigned"] and m["signed"]:
        fl = m["flushloop"]
        print("%-46s %10.4f %10s %10.4f %7s %7.3f" % (k[:46], m["unsigned"], "%.4f" % fl if fl else "-", m["signed"],
                                                     "%.3f" % (fl / m["unsigned"]) if fl else "-", m["signed"] / m["unsigned"]))
