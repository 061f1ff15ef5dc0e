#!/bin/bash
# rocprofv3 kernel stats of the small-n path: tools/prof_small.sh <out-dir> <logn> [curve group]
out=$(realpath -m $1); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python /root/repo/tools/sweep_small.py ${3:-bn254} ${4:-g1} --logns=$2 > $out/run.log 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
grep "^2\^" $out/run.log
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls']) >= 40: print(f"{r['Name'][:72]:72} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
