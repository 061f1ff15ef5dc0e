#!/bin/bash
# k_part_scatter: write-out through a per-partition displacement in LDS (shipped) against the two global look-ups per
# entry (build_ab_prescat = the previous commit, BN254 G1): parity subset first, then stage times, alternated twice.
S=${1:-s12}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or all_c or random_matches or edge or skewed or baseline_config" > $O/gputest_subset.log 2>&1; echo "pytest rc=$?" >> $O/gputest_subset.log ); tail -3 $O/gputest_subset.log
AB=/root/repo/gnark-crypto_amd/csrc/build_ab_prescat/libgmsm_ab.so
for rep in 1 2; do
for logn in 20 22 24 26; do
  echo "== 2^$logn displacement in LDS (shipped), run $rep"; python tools/sweep_env.py bn254 g1 $logn 10 -- "" "" 2>&1 | tail -1
  echo "== 2^$logn global look-ups, run $rep"; GMSM_LIB=$AB python tools/sweep_env.py bn254 g1 $logn 10 -- "" "" 2>&1 | tail -1
done
done > $O/scatter_writeout_ab.log 2>&1
cat $O/scatter_writeout_ab.log
