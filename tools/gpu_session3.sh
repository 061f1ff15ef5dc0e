#!/bin/bash
# Session 3: the Cooley-Tukey-everywhere FFT: parity suite, A/B against the frozen round-3 kernels (build_ab_fft0), the
# other fields, and the rocprofv3 trace + PMC passes of the new kernels.
S=${1:-s3}
cd /root/repo
O=gpurun_out/$S
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fft.py -x -q > $O/fft_tests.log 2>&1; echo "pytest rc=$?" >> $O/fft_tests.log ); tail -3 $O/fft_tests.log
for v in default nofork fft0 default nofork fft0; do
  if [ $v = default ]; then unset GMSM_LIB; else export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_$v/libgmsm_ab.so; fi
  echo "== $v"; timeout 300 python tools/bench_fft.py bn254 16 20 22 24
done > $O/fft_ab.log 2>&1
unset GMSM_LIB
( timeout 200 python tools/bench_fft.py bls12_381 20 24; timeout 200 python tools/bench_fft.py bw6_761 20 24 ) > $O/fft_other.log 2>&1
grep -E "==|2\^24|2\^20" $O/fft_ab.log; grep "2\^24" $O/fft_other.log
tools/profile_fft.sh $S/fft_prof bn254 20 24
