#!/bin/bash
mkdir -p gpurun_out/r2h
cd /root/repo
B=tools/bench_short.sh
L=$PWD/gnark-crypto_amd/csrc/build_ab
{
$B new_blsg2_20 --curve bls12_381 --group g2 --logn 20 --steps 5
GMSM_LIB=$L/libgmsm_inl.so $B inl_blsg2_20 --curve bls12_381 --group g2 --logn 20 --steps 5
$B new_bw6_18 --curve bw6_761 --group g1 --logn 18 --steps 5
GMSM_LIB=$L/libgmsm_inl.so $B inl_bw6_18 --curve bw6_761 --group g1 --logn 18 --steps 5
$B new_bw6_20 --curve bw6_761 --group g1 --logn 20 --steps 3
GMSM_LIB=$L/libgmsm_inl.so $B inl_bw6_20 --curve bw6_761 --group g1 --logn 20 --steps 3
$B new_blsg2_22 --curve bls12_381 --group g2 --logn 22 --steps 3
GMSM_LIB=$L/libgmsm_inl.so $B inl_blsg2_22 --curve bls12_381 --group g2 --logn 22 --steps 3
} > gpurun_out/r2h/ab.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bls12_381 or bw6_761" 2>&1 | tail -4) > gpurun_out/r2h/pytest.log
