#!/bin/bash
mkdir -p gpurun_out/r2c
cd /root/repo
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r2c/pytest.log
B=tools/bench_short.sh
L=$PWD/gnark-crypto_amd/csrc/build_ab
{
$B default
$B default_again
GMSM_LIB=$L/libgmsm_e2.so $B exp2_nogather
GMSM_LIB=$L/libgmsm_e7.so $B exp7_all
$B default_22 --logn 22
$B default_24 --logn 24 --steps 5
GMSM_LIB=$L/libgmsm_e2.so $B exp2_24 --logn 24 --steps 5
GMSM_LIB=$L/libgmsm_e7.so $B exp7_24 --logn 24 --steps 5
} > gpurun_out/r2c/ab.log 2>&1
