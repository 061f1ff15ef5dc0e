set -x
timeout 900 python -m pytest tests/test_gpu_fft.py -x -q -m gpu 2>&1 | tail -5
for cv in bn254 bls12_381 bw6_761; do
python tools/bench_fft.py $cv 16 20 22 24 2>&1 | grep fft
GMSM_FFT_LAZY=0 python tools/bench_fft.py $cv 20 24 2>&1 | grep fft | sed 's/^/SATURATED /'
done
