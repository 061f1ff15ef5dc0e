#!/bin/bash
# round-2 GPU call A: parity, split/X2 A-B, ubench
mkdir -p gpurun_out/r2a
cd /root/repo
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r2a/pytest.log
B=tools/bench_short.sh
{
$B default
GMSM_SPLIT=1 $B split1
GMSM_SPLIT=2 $B split2
GMSM_SPLIT=2 GMSM_SPLIT_LAST=6 $B split2_last6
GMSM_SPLIT=2 GMSM_SPLIT_LAST=8 $B split2_last8
GMSM_SPLIT=3 GMSM_SPLIT_LAST=2 $B split3_last2
GMSM_SPLIT=3 GMSM_SPLIT_LAST=6 $B split3_last6
GMSM_SPLIT=4 GMSM_SPLIT_LAST=2 $B split4_last2
GMSM_SPLIT=4 GMSM_SPLIT_LAST=4 $B split4_last4
GMSM_DIGIT32=1 $B digit32
GMSM_LIB=$PWD/gnark-crypto_amd/csrc/build_ab/libgmsm_x1.so GMSM_SPLIT=1 $B x1_split1
GMSM_LIB=$PWD/gnark-crypto_amd/csrc/build_ab/libgmsm_x1.so $B x1_default
for ln in 16 18 22; do $B default_$ln --logn $ln; GMSM_SPLIT=1 $B split1_$ln --logn $ln; done
$B default_24 --logn 24 --steps 5
GMSM_SPLIT=1 $B split1_24 --logn 24 --steps 5
GMSM_SPLIT=2 GMSM_SPLIT_LAST=8 $B split2h_24 --logn 24 --steps 5
} > gpurun_out/r2a/ab.log 2>&1
tools/ubench_fpmul > gpurun_out/r2a/ubench_fpmul.log 2>&1
python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err
