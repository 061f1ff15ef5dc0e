#!/bin/bash
# Same-box A/B of k_reduce_serial_q's workgroup shape (tools/build_ab.sh -DGMSM_SERIAL_Q_QUADS=16|32 [-DGMSM_SERIAL_Q_WAVES=2]): the shipped
# library (64 quads per workgroup) against build_ab_q16 / q32 / q16w2, stage times of the groups whose serial reduction runs on quads.
cd /root/repo
C=gnark-crypto_amd/csrc
for cfg in "bw6_761 g1 20" "bls12_381 g2 22" "bn254 g2 20" "bls12_381 g1 22"; do
  for lib in $C/libgmsm.so $C/build_ab_q16/libgmsm_ab.so $C/build_ab_q32/libgmsm_ab.so $C/build_ab_q16w2/libgmsm_ab.so; do
    [ -f $lib ] || continue
    echo "## $cfg  $lib"
    GMSM_LIB=$PWD/$lib timeout 300 python tools/glv_ab.py $cfg --steps=6 2>&1 | grep "^c="
  done
done
