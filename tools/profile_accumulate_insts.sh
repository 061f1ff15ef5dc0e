#!/bin/bash
# Instruction counts of k_accumulate_seg per launch (rocprofv3 PMC, no trace flags), BN254 G1 2^20, for the shipped
# signed-limb form and for the unsigned form (build_ab_unsigned: -DGMSM_SIGNED_MADD=0) -> gpurun_out/$1/
out=/root/repo/gpurun_out/${1:-acc_insts}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-entry --no-pipeline --no-also --no-next-rows --steps 3 --warmup 1"
for kind in signed unsigned; do
  if [ $kind = unsigned ]; then export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_unsigned/libgmsm_ab.so; else unset GMSM_LIB; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $out/${kind}_insts -o p --output-format csv -- python /root/repo/bench.py $Q > /dev/null 2> $out/${kind}_insts.log
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $out/${kind}_time -o p --output-format csv -- python /root/repo/bench.py $Q > /dev/null 2> $out/${kind}_time.log
done
find $out -name "*agent_info.csv" -delete
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for kind in ("signed", "unsigned"):
    for p in ("insts", "time"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for f in glob.glob(f"{out}/{kind}_{p}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "k_accumulate_seg" not in k: continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] in ("SQ_WAVES", "SQ_WAVE_CYCLES"): n[k] += 1
        for k, v in acc.items():
            print(kind, p, "launches", n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
PY
