#!/bin/bash
# Fp2 mixed addition on signed limbs (madd_ts, shipped) against build_ab_fp2unsigned (-DGMSM_SIGNED_MADD2=0: madd_g for
# BN254 G2, madd_t for BLS12-381 G2): the G2 parity tests first, then stage times, alternated twice; fuzz on the G2 groups.
S=${1:-s9}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q -k "g2 or G2 or parity or lazy or group" > $O/gputest_g2.log 2>&1; echo "pytest rc=$?" >> $O/gputest_g2.log ); tail -3 $O/gputest_g2.log
AB=/root/repo/gnark-crypto_amd/csrc/build_ab_fp2unsigned/libgmsm_ab.so
for rep in 1 2; do
for cfg in "bn254 g2 20" "bn254 g2 22" "bls12_381 g2 20" "bls12_381 g2 22"; do
  echo "== $cfg signed (shipped), run $rep"; python tools/sweep_env.py $cfg 5 -- "" "" 2>&1 | tail -1
  echo "== $cfg unsigned forms, run $rep"; GMSM_LIB=$AB python tools/sweep_env.py $cfg 5 -- "" "" 2>&1 | tail -1
done
done > $O/fp2_signed_ab.log 2>&1
cat $O/fp2_signed_ab.log
for grp in "bn254 g2" "bls12_381 g2" "bls12_381 g1" "bw6_761 g1" "bw6_761 g2"; do timeout 200 python tools/fuzz_parity.py 30 $grp 2>&1 | tail -1; done > $O/fuzz.log 2>&1
cat $O/fuzz.log
