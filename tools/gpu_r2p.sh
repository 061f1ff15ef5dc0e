#!/bin/bash
# last GPU call of round 2 (5.8 GPU-minutes left): canary for the c > 16 small-n geometry, full GPU suite, bench, and -
# only if time is left - the UnsatOpsMid A/B library on BW6-761 G1 and BN254 G2. Every step has its own short timeout.
out=/root/repo/gpurun_out/r2p
mkdir -p $out
cd /root/repo
timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -k "test_msm_sum_of_squares_identity_all_c and bn254-g1" > $out/canary.log 2>&1
rc=$?; echo "canary exit $rc"; tail -2 $out/canary.log
timeout 75 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
if [ $rc -eq 0 ]; then
  timeout 175 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" | tee -a $out/pytest.log
else
  timeout 175 python -m pytest tests -m gpu -x -q -k "not sum_of_squares and not edge_cases" > $out/pytest.log 2>&1; echo "pytest (reduced) exit $?" | tee -a $out/pytest.log
fi
tail -3 $out/pytest.log
MID=/root/repo/gnark-crypto_amd/csrc/build_ab/libgmsm_mid.so
if [ -f $MID ]; then
  GMSM_LIB=$MID timeout 30 python tools/sweep_env.py bw6_761 g1 20 2 -- "" "GMSM_C=16" > $out/mid_ab.log 2>&1
  GMSM_LIB=$MID timeout 20 python tools/sweep_env.py bn254 g2 20 3 -- "" >> $out/mid_ab.log 2>&1
  timeout 20 python tools/sweep_env.py bw6_761 g1 20 2 -- "" "GMSM_C=16" >> $out/mid_ab.log 2>&1
  grep -v amdgpu $out/mid_ab.log
fi
