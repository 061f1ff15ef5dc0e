#!/bin/bash
# large-n window widths for BN254 G1 (c > 16 with the split reduction) and per-size window sweeps of the wide groups
out=/root/repo/gpurun_out/r2n
mkdir -p $out
cd /root/repo
V=("" "GMSM_C=17" "GMSM_C=17,GMSM_SPLIT_REDUCE=1" "GMSM_C=20" "GMSM_C=20,GMSM_SPLIT_REDUCE=1" \
   "GMSM_C=20,GMSM_SPLIT_REDUCE=1,GMSM_LOG2L=5" "GMSM_C=20,GMSM_SPLIT_REDUCE=1,GMSM_LOG2L=6" "GMSM_C=20,GMSM_SPLIT_REDUCE=1,GMSM_LOG2L=7" \
   "GMSM_C=22" "GMSM_C=22,GMSM_SPLIT_REDUCE=1" "GMSM_C=20,GMSM_PART_LOG2=14" "GMSM_C=20,GMSM_PART_LOG2=16" "GMSM_SPLIT_REDUCE=1")
timeout 300 python tools/sweep_env.py bn254 g1 22 4 -- "${V[@]}" > $out/large_c_22.log 2>&1
timeout 300 python tools/sweep_env.py bn254 g1 24 4 -- "${V[@]}" > $out/large_c_24.log 2>&1
timeout 400 python tools/sweep_env.py bn254 g1 26 3 -- "${V[@]}" > $out/large_c_26.log 2>&1
timeout 300 python tools/sweep_c.py 10 19 bw6_761 g1 8 16 > $out/sweep_bw6_g1.log 2>&1
timeout 300 python tools/sweep_c.py 10 19 bw6_761 g2 8 16 > $out/sweep_bw6_g2.log 2>&1
timeout 300 python tools/sweep_c.py 10 21 bls12_381 g2 8 16 > $out/sweep_bls_g2.log 2>&1
timeout 300 python tools/sweep_c.py 10 21 bls12_381 g1 8 16 > $out/sweep_bls_g1.log 2>&1
timeout 300 python tools/sweep_c.py 10 19 bn254 g2 8 16 > $out/sweep_bn_g2.log 2>&1
grep -h -v amdgpu.ids $out/*.log
