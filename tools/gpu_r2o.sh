#!/bin/bash
# measured window table + UnsatOpsMid for the wide groups: GPU suite, bench, large-n widths of the other groups
out=/root/repo/gpurun_out/r2o
mkdir -p $out
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
V=("" "GMSM_C=14" "GMSM_C=16" "GMSM_C=17")
timeout 300 python tools/sweep_env.py bn254 g1 21 4 -- "" "GMSM_C=17" > $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bn254 g2 22 3 -- "${V[@]}" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bls12_381 g1 22 3 -- "${V[@]}" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bls12_381 g1 24 3 -- "${V[@]}" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bls12_381 g2 22 3 -- "${V[@]}" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bw6_761 g1 20 3 -- "" "GMSM_C=9" "GMSM_C=12" "GMSM_C=13" "GMSM_C=14" "GMSM_C=16" "GMSM_C=14,GMSM_LOG2L=2" "GMSM_C=14,GMSM_LOG2L=3" "GMSM_C=14,GMSM_LOG2L=4" "GMSM_C=16,GMSM_LOG2L=3" "GMSM_C=16,GMSM_LOG2L=4" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bw6_761 g1 22 2 -- "" "GMSM_C=14" "GMSM_C=16" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bw6_761 g1 16 4 -- "" "GMSM_C=12" "GMSM_C=14" >> $out/more_c.log 2>&1
timeout 300 python tools/sweep_env.py bls12_381 g2 16 4 -- "" "GMSM_C=10" "GMSM_C=13" "GMSM_C=16" >> $out/more_c.log 2>&1
grep -h -v amdgpu.ids $out/more_c.log
