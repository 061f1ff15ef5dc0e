#!/bin/bash
# One GPU-box session: tools/gpu_session.sh <label> <step> [<step> ...]  ->  gpurun_out/<label>/
# (one parametrised script instead of a numbered script per session). Steps:
#   suite        pytest -m gpu (GMSM_REQUIRE_SANITIZERS=1: the sanitizer tests fail instead of skipping)
#   smoke        __graft_entry__.smoke()
#   bench        python bench.py (default line)
#   dist         tools/bench_distributions.py (BN254 G1 2^20 + 2^24, all distributions)
#   dist_g2      ... BLS12-381 G2 2^22, uniform + smallvalues
#   dist_prof    rocprofv3 kernel stats of the smallvalues rows (2^20, 2^24)
#   small        tools/bench_small_n.py (BN254 G1)
#   small_all    ... the other five groups, no CPU legs
#   profile      tools/profile_round.sh (kernel traces + PMC traffic per BASELINE configuration)
#   cmd:<shell>  any command line (quoted), logged as cmd_<k>.log
L=${1:?label}; shift
cd /root/repo
O=gpurun_out/$L; mkdir -p $O
k=0
for step in "$@"; do
  t0=$(date +%s)
  case "$step" in
    suite)   ( GMSM_REQUIRE_SANITIZERS=1 timeout 2400 python -m pytest tests -m gpu -x -q -rs > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -4 $O/gputest.log
             cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench)   ( timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); tail -c 900 $O/bench.json; echo; tail -2 $O/bench.err ;;
    dist)    timeout 900 python tools/bench_distributions.py > $O/dist.json 2> $O/dist.log; cat $O/dist.log | grep -v Warning ;;
    dist_g2) timeout 900 python tools/bench_distributions.py bls12_381 g2 22 3 --kinds=uniform,smallvalues --no-cold > $O/dist_g2.json 2> $O/dist_g2.log; grep -v Warning $O/dist_g2.log ;;
    dist_prof)
             ( cd /tmp && export TMPDIR=/tmp
               for logn in 20 24; do
                 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_small_$logn -o t --output-format csv -- \
                   python /root/repo/tools/bench_distributions.py bn254 g1 $logn 3 --kinds=smallvalues --no-cold > /dev/null 2> /root/repo/$O/prof_small_$logn.log
                 find /root/repo/$O/prof_small_$logn -name "*kernel_trace.csv" -delete; find /root/repo/$O/prof_small_$logn -name "*agent_info.csv" -delete
                 f=$(find /root/repo/$O/prof_small_$logn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
               done ) ;;
    small)   timeout 900 python tools/bench_small_n.py > $O/small_bn254_g1.json 2> $O/small_bn254_g1.log; grep -v Warning $O/small_bn254_g1.log ;;
    small_all)
             for cg in "bn254 g2" "bls12_381 g1" "bls12_381 g2" "bw6_761 g1"; do
               set -- $cg; timeout 600 python tools/bench_small_n.py $1 $2 --no-cpu --logns=5,8,10,12 > $O/small_$1_$2.json 2> $O/small_$1_$2.log; echo "== $cg"; grep -v Warning $O/small_$1_$2.log
             done ;;
    profile) timeout 1500 tools/profile_round.sh $L/prof > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log ;;
    cmd:*)   k=$((k+1)); ( timeout ${GMSM_CMD_TIMEOUT:-600} bash -c "${step#cmd:}" ) > $O/cmd_$k.log 2>&1; echo "[cmd_$k rc=$?]"; tail -30 $O/cmd_$k.log ;;
    *)       echo "unknown step $step" ;;
  esac
  echo "[$step: $(( $(date +%s) - t0 )) s]"
done
