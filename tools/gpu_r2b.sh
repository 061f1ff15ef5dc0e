#!/bin/bash
mkdir -p gpurun_out/r2b
cd /root/repo
B=tools/bench_short.sh
L=$PWD/gnark-crypto_amd/csrc/build_ab
{
$B default
GMSM_LIB=$L/libgmsm_e1.so $B exp1_noflush
GMSM_LIB=$L/libgmsm_e2.so $B exp2_nogather
GMSM_LIB=$L/libgmsm_e4.so $B exp4_nobound
GMSM_LIB=$L/libgmsm_e7.so $B exp7_all
GMSM_SEG=64 $B seg64
GMSM_SEG=128 $B seg128
GMSM_SEG=170 $B seg170
$B default_24 --logn 24 --steps 5
GMSM_LIB=$L/libgmsm_e7.so $B exp7_24 --logn 24 --steps 5
GMSM_LIB=$L/libgmsm_e2.so $B exp2_24 --logn 24 --steps 5
} > gpurun_out/r2b/ab.log 2>&1
tools/ubench_madd > gpurun_out/r2b/ubench_madd.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d /root/repo/gpurun_out/r2b/pmc_sq -o sq --output-format csv -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry --no-pipeline > /root/repo/gpurun_out/r2b/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d /root/repo/gpurun_out/r2b/pmc_sq2 -o sq2 --output-format csv -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry --no-pipeline > /root/repo/gpurun_out/r2b/pmc_sq2.log 2>&1
