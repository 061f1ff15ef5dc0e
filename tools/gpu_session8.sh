#!/bin/bash
# k_part_scatter without its counting pass (populations from the chunk prefixes) against the build that counts
# (build_ab_scattercount: -DGMSM_SCATTER_COUNT=1, BN254 G1), stage times at 2^20..2^26; the GPU suite and the fuzz on the
# shipped build (signed limbs + this).
S=${1:-s8}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null
AB=/root/repo/gnark-crypto_amd/csrc/build_ab_scattercount/libgmsm_ab.so
for rep in 1 2; do
for logn in 20 22 24 26; do
  echo "== 2^$logn shipped (no counting pass), run $rep"; python tools/sweep_env.py bn254 g1 $logn 10 -- "" "" 2>&1 | tail -2
  echo "== 2^$logn counting pass, run $rep"; GMSM_LIB=$AB python tools/sweep_env.py bn254 g1 $logn 10 -- "" "" 2>&1 | tail -2
done
done > $O/scatter_ab.log 2>&1
cat $O/scatter_ab.log
for grp in "bn254 g1" "bls12_381 g1" "bw6_761 g1" "bw6_761 g2"; do timeout 200 python tools/fuzz_parity.py 40 $grp 2>&1 | tail -2; done > $O/fuzz.log 2>&1
cat $O/fuzz.log
