#!/bin/bash
# Reduction geometry and window width of the single BN254 G1 call at 2^18..2^21, forced in an experiments build.
S=${1:-s6}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_exp/libgmsm_ab.so
for logn in 20 18 21; do
python tools/sweep_env.py bn254 g1 $logn 10 -- "" "GMSM_LOG2L=2" "GMSM_LOG2L=3" "GMSM_LOG2L=4" "GMSM_REDUCE_LEVELS=3,GMSM_LOG2L=1" "GMSM_REDUCE_LEVELS=3,GMSM_LOG2L=2" "GMSM_REDUCE_LEVELS=3,GMSM_LOG2L=3" "GMSM_C=15" "GMSM_C=15,GMSM_LOG2L=2" "GMSM_C=15,GMSM_REDUCE_LEVELS=3,GMSM_LOG2L=2" "GMSM_C=14" ""
done > $O/reduce_sweep.log 2>&1
cat $O/reduce_sweep.log
