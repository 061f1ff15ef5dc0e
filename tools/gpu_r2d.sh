#!/bin/bash
mkdir -p gpurun_out/r2d
cd /root/repo
B=tools/bench_short.sh
L=$PWD/gnark-crypto_amd/csrc/build_ab
{
$B new
GMSM_LIB=$L/libgmsm_old.so $B old
GMSM_COOP=1 $B coop
$B new2
GMSM_LIB=$L/libgmsm_old.so $B old2
GMSM_COOP=1 $B coop2
$B new_24 --logn 24 --steps 5
GMSM_LIB=$L/libgmsm_old.so $B old_24 --logn 24 --steps 5
GMSM_COOP=1 $B coop_24 --logn 24 --steps 5
$B new_22 --logn 22
GMSM_COOP=1 $B coop_22 --logn 22
} > gpurun_out/r2d/ab.log 2>&1
GMSM_COOP=1 timeout 600 python -m pytest tests -m gpu -x -q -k "multiexp or msm or window" 2>&1 | tail -3 > gpurun_out/r2d/pytest_coop.log
