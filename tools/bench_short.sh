#!/bin/bash
# usage: tools/bench_short.sh [label] [bench args...]  -> one compact line with the per-stage times
label=$1; shift
python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
