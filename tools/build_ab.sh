#!/bin/bash
# A/B library for the wide element types: gnark-crypto_amd/csrc/$D/libgmsm_ab.so = the shipped objects with groups
# 1 (BN254 G2), 3 (BLS12-381 G2), 4 (BW6-761 G1) rebuilt with the given flags. Run on the CPU box before a gpurun call
# (the library travels with the snapshot); load it with GMSM_LIB=<path>.
# AB_NAME=x puts the library into build_ab_x/.
# usage: tools/build_ab.sh [-DGMSM_COMBINE_LDS=1 -DGMSM_FIXLONG_INLINE=1 ...]   (default: those two)
set -e
cd "$(dirname "$0")/../gnark-crypto_amd/csrc"
FLAGS="${*:--DGMSM_COMBINE_LDS=1 -DGMSM_FIXLONG_INLINE=1}"
make -j"$(nproc)" libgmsm.so > /dev/null
D=build_ab${AB_NAME:+_$AB_NAME}
mkdir -p $D
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function"
for g in 1 3 4; do
  /opt/rocm/bin/hipcc $F $FLAGS -DGMSM_GROUP_ID=$g -c -o $D/group$g.o gmsm_group_inst.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgmsm_ab.so build/engine.o build/group0.o $D/group1.o \
  build/group2.o $D/group3.o $D/group4.o build/group5.o
echo "built $(pwd)/$D/libgmsm_ab.so with: $FLAGS"
