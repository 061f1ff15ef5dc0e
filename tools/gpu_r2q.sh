#!/bin/bash
# A/B of the inlined level-1 combine (k_reduce_combine<UnsatOps>, -DGMSM_COMBINE_INLINE=1 for BN254 G2, BLS12-381 G2,
# BW6-761 G1) against the shipped library, then a subset of the parity suite on the A/B library. 2.9 GPU-minutes left.
out=/root/repo/gpurun_out/r2q
mkdir -p $out
cd /root/repo
AB=/root/repo/gnark-crypto_amd/csrc/build_ab/libgmsm_ci.so
GMSM_LIB=$AB timeout 25 python tools/sweep_env.py bw6_761 g1 20 2 -- "" "GMSM_C=16" "GMSM_LOG2L=4" "GMSM_LOG2L=2" > $out/ci_ab.log 2>&1; echo "bw6 exit $?"
GMSM_LIB=$AB timeout 25 python tools/sweep_env.py bls12_381 g2 20 2 -- "" "GMSM_C=13" "GMSM_LOG2L=3" >> $out/ci_ab.log 2>&1; echo "bls g2 exit $?"
GMSM_LIB=$AB timeout 20 python tools/sweep_env.py bn254 g2 20 3 -- "" "GMSM_C=15" >> $out/ci_ab.log 2>&1; echo "bn g2 exit $?"
grep -v amdgpu $out/ci_ab.log
GMSM_LIB=$AB timeout 75 python -m pytest tests/test_gpu_parity.py -x -q -k "(bw6_761-g1 or bls12_381-g2 or bn254-g2) and (sum_of_squares or edge_cases or random or skewed or baseline)" > $out/pytest_ab.log 2>&1; echo "pytest exit $?"
tail -3 $out/pytest_ab.log
