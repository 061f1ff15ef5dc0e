#!/bin/bash
# A/B of tools/build_ab.sh's library against the shipped one on the wide groups (about 1 GPU-minute): per-stage times at
# the BASELINE sizes, then the parity tests of those groups on the A/B library. Every step has its own timeout.
out=/root/repo/gpurun_out/ab_wide
mkdir -p $out
cd /root/repo
AB=/root/repo/gnark-crypto_amd/csrc/build_ab/libgmsm_ab.so
[ -f $AB ] || { echo "run tools/build_ab.sh first"; exit 1; }
for lib in "" $AB; do
  echo "== library: ${lib:-shipped}" >> $out/ab.log
  GMSM_LIB=$lib timeout 40 python tools/sweep_env.py bw6_761 g1 20 3 -- "" "GMSM_C=16" >> $out/ab.log 2>&1
  GMSM_LIB=$lib timeout 40 python tools/sweep_env.py bls12_381 g2 20 3 -- "" >> $out/ab.log 2>&1
  GMSM_LIB=$lib timeout 30 python tools/sweep_env.py bn254 g2 20 3 -- "" >> $out/ab.log 2>&1
done
grep -v amdgpu $out/ab.log
GMSM_LIB=$AB timeout 90 python -m pytest tests/test_gpu_parity.py -x -q -k "(bw6_761-g1 or bls12_381-g2 or bn254-g2) and (sum_of_squares or edge_cases or random or skewed or baseline)" > $out/pytest_ab.log 2>&1
echo "pytest (A/B library) exit $?"; tail -3 $out/pytest_ab.log
