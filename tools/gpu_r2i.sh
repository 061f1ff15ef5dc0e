#!/bin/bash
mkdir -p gpurun_out/r2i
cd /root/repo
B=tools/bench_short.sh
L=$PWD/gnark-crypto_amd/csrc/build_ab
{
for v in n1 n2 n4; do
  if [ $v = n1 ]; then unset GMSM_LIB; else export GMSM_LIB=$L/libgmsm_$v.so; fi
  $B ${v}_blsg2_20 --curve bls12_381 --group g2 --logn 20 --steps 5
  $B ${v}_bw6_18 --curve bw6_761 --group g1 --logn 18 --steps 5
  $B ${v}_bw6_20 --curve bw6_761 --group g1 --logn 20 --steps 3
  $B ${v}_blsg2_22 --curve bls12_381 --group g2 --logn 22 --steps 3
done
} > gpurun_out/r2i/ab.log 2>&1
export GMSM_LIB=$L/libgmsm_n2.so
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bls12_381 or bw6_761" 2>&1 | tail -4) > gpurun_out/r2i/pytest_n2.log
export GMSM_LIB=$L/libgmsm_n4.so
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bls12_381 or bw6_761" 2>&1 | tail -4) > gpurun_out/r2i/pytest_n4.log
