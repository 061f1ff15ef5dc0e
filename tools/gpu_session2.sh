#!/bin/bash
# Session 2: fr/fft A/B of the round-4 changes (same box, same call), the FFT tests on the new default, the batch-affine
# prototype with a full window set's worth of pairs.
S=${1:-s2}
cd /root/repo
O=gpurun_out/$S
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q > $O/fft_tests.log 2>&1; echo "pytest rc=$?" >> $O/fft_tests.log ); tail -3 $O/fft_tests.log
for i in 1 2; do timeout 600 python bench.py --no-also --no-cpu-baseline --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^20', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items()}, d['value_cold'], d['value_warm_bases'])"; done
timeout 600 python bench.py --logn 24 --steps 5 --no-also --no-cpu-baseline --no-next-rows --no-host-entry --no-pipeline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^24', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items()})"
for v in default fft0 fft1 fft2; do
  if [ $v = default ]; then unset GMSM_LIB; else export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_$v/libgmsm_ab.so; fi
  [ $v != default ] && [ ! -f "$GMSM_LIB" ] && continue
  for rep in 1 2; do
    echo "== $v (rep $rep)"; timeout 300 python tools/bench_fft.py bn254 16 20 22 24
  done
done > $O/fft_ab.log 2>&1
unset GMSM_LIB
( timeout 200 python tools/bench_fft.py bls12_381 20 24; timeout 200 python tools/bench_fft.py bw6_761 20 24 ) > $O/fft_other.log 2>&1
grep -E "==|2\^24" $O/fft_ab.log
( timeout 300 tools/ubench_batch_affine 20 26; timeout 300 tools/ubench_batch_affine 24 26 ) > $O/batch_affine.log 2>&1
cat $O/batch_affine.log
tools/profile_fft.sh $S/fft_prof bn254 24
