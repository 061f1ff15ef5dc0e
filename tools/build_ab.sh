#!/bin/bash
# A/B library: gnark-crypto_amd/csrc/build_ab[_NAME]/libgmsm_ab.so = the shipped objects with the groups of AB_GROUPS
# (default "1 3 4": BN254 G2, BLS12-381 G2, BW6-761 G1) rebuilt with the given flags, e.g. -DGMSM_EXPERIMENTS=1 (the
# tuning knobs become run-time GMSM_* switches), -DGMSM_SERIAL_QUAD_WORDS=9, -DGMSM_COMBINE_WE_WORDS=0. Run on the CPU box
# before a gpurun call (the library travels with the snapshot); load it with GMSM_LIB=<path>.
# AB_NAME=x puts the library into build_ab_x/.
# usage: [AB_NAME=x] [AB_GROUPS="0 2"] tools/build_ab.sh -DFLAG=1 ...
set -e
cd "$(dirname "$0")/../gnark-crypto_amd/csrc"
FLAGS="${*:--DGMSM_EXPERIMENTS=1}"
GROUPS_AB="${AB_GROUPS:-1 3 4}"
make -j"$(nproc)" libgmsm.so > /dev/null
D=build_ab${AB_NAME:+_$AB_NAME}
mkdir -p $D
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function"
for g in $GROUPS_AB; do
  /opt/rocm/bin/hipcc $F $FLAGS -DGMSM_GROUP_ID=$g -c -o $D/group$g.o gmsm_group_inst.hip &
done
wait
OBJS=""
for g in 0 1 2 3 4 5; do
  if [[ " $GROUPS_AB " == *" $g "* ]]; then OBJS="$OBJS $D/group$g.o"; else OBJS="$OBJS build/group$g.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgmsm_ab.so build/engine.o $OBJS
echo "built $(pwd)/$D/libgmsm_ab.so with: $FLAGS"
