#!/bin/bash
mkdir -p /root/repo/gpurun_out/r2k
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-entry --no-pipeline --no-also --curve bw6_761 --group g1 --logn 18 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d /root/repo/gpurun_out/r2k/sq1 -o a --output-format csv -- python /root/repo/bench.py $Q > /dev/null 2> /root/repo/gpurun_out/r2k/sq1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT -d /root/repo/gpurun_out/r2k/sq2 -o b --output-format csv -- python /root/repo/bench.py $Q > /dev/null 2> /root/repo/gpurun_out/r2k/sq2.log
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH GRBM_GUI_ACTIVE GRBM_COUNT -d /root/repo/gpurun_out/r2k/sq3 -o c --output-format csv -- python /root/repo/bench.py $Q > /dev/null 2> /root/repo/gpurun_out/r2k/sq3.log
cd /root/repo
python - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/r2k/sq*/*counter_collection.csv')):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_accumulate_seg' in k or 'k_reduce1' in k:
            d[k.split('<')[0][-20:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items():
        print(f, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
