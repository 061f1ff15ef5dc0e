#!/bin/bash
# Closing session of round 4 on the final build: GPU suite, smoke(), stage times (histogram with 16-byte loads), the
# per-configuration rocprofv3 summaries (kernel trace + FETCH / WRITE passes) and the bench line.
S=${1:-s11}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for logn in 20 22 24; do python tools/sweep_env.py bn254 g1 $logn 10 -- "" "" 2>&1 | tail -1; done > $O/stages.log 2>&1; cat $O/stages.log
tools/profile_round.sh $S/prof > $O/profile_round.log 2>&1
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); tail -c 700 $O/bench.json; echo
