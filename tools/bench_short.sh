#!/bin/bash
# usage: tools/bench_short.sh [label] [bench args...]  -> one compact line with the per-stage times
label=$1; shift
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --no-pipeline --no-also "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()}, 'acc_launch_ms', round(d['roofline']['avg_launch_ms'],3), 'x', d['roofline'].get('launches_per_msm'))"
