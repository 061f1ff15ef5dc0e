#!/bin/bash
# VGPRs / scratch / LDS of every kernel in a built group object: tools/kernel_resources.sh build/group4.o [filter]
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$1 --output=$tmp/dev.o --unbundle 2>/dev/null \
  || { objcopy -O binary --only-section=.hip_fatbin $1 $tmp/fat.bin; $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.o --unbundle; }
$LLVM/llvm-readelf --notes $tmp/dev.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk); v=re.search(r'\.vgpr_count:\s+(\d+)',blk); sc=re.search(r'\.private_segment_fixed_size:\s+(\d+)',blk); l=re.search(r'\.group_segment_fixed_size:\s+(\d+)',blk); sp=re.search(r'\.vgpr_spill_count:\s+(\d+)',blk)
    print(f'{v.group(1):>4} vgpr {sc.group(1):>6} scratch {(sp.group(1) if sp else \"-\"):>5} spill {l.group(1):>6} lds  {name.group(1)[:110]}')
" | (if [ -n "$2" ]; then grep -E "$2"; else cat; fi)
rm -rf $tmp
