#!/bin/bash
# re-entry check of round 2: GPU suite, the default bench line, window-size sweeps for the wide groups
out=/root/repo/gpurun_out/r2m
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
timeout 300 python tools/sweep_c.py 20 20 bw6_761 g1 10 16 > $out/sweep_bw6.log 2>&1
timeout 300 python tools/sweep_c.py 22 22 bls12_381 g2 11 16 > $out/sweep_bls_g2.log 2>&1
timeout 300 python tools/sweep_c.py 22 22 bls12_381 g1 12 16 > $out/sweep_bls_g1.log 2>&1
timeout 300 python tools/sweep_c.py 20 20 bn254 g2 11 16 > $out/sweep_bn_g2.log 2>&1
timeout 300 python tools/sweep_c.py 18 18 bw6_761 g1 10 16 >> $out/sweep_bw6.log 2>&1
cat $out/sweep_*.log
