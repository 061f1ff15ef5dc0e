set -x
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bls12_381 or group_op or field" 2>&1 | tail -4
tools/bench_short.sh "bls381g2-2^22 tracked" --curve bls12_381 --group g2 --logn 22 --steps 5
tools/bench_short.sh "bls381g2-2^20 tracked" --curve bls12_381 --group g2 --logn 20 --steps 5
