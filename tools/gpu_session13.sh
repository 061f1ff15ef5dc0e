#!/bin/bash
# Closing session on the final build: GPU suite, smoke(), the reduction's stage time with the bucket prefetch (shipped)
# against build_ab_prescat (two commits earlier: no prefetch), the rocprofv3 summaries and the bench line.
S=${1:-s13}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
AB=/root/repo/gnark-crypto_amd/csrc/build_ab_prescat/libgmsm_ab.so
for rep in 1 2; do
for logn in 16 18 20; do
  echo "== 2^$logn bucket prefetch (shipped), run $rep"; python tools/sweep_env.py bn254 g1 $logn 20 -- "" "" 2>&1 | tail -1
  echo "== 2^$logn no prefetch, run $rep"; GMSM_LIB=$AB python tools/sweep_env.py bn254 g1 $logn 20 -- "" "" 2>&1 | tail -1
done
done > $O/reduce_prefetch_ab.log 2>&1
cat $O/reduce_prefetch_ab.log
tools/profile_round.sh $S/prof > $O/profile_round.log 2>&1
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); tail -c 600 $O/bench.json; echo
