#!/bin/bash
# rocprofv3 kernel stats of one distribution row: tools/prof_dist.sh <out-dir> <curve> <group> <logn> <kind> [steps] [--glv=N]
out=$(realpath -m $1); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python /root/repo/tools/bench_distributions.py $2 $3 $4 ${6:-3} --kinds=$5 --no-cold ${7:-} > /dev/null 2> $out/run.log
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:64]:64} calls {r['Calls']:>4} avg_us {float(r['AverageNs'])/1e3:10.1f} max_us {float(r['MaxNs'])/1e3:10.1f}")
PY
