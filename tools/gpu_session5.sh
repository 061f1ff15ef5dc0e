#!/bin/bash
# Why the library's HIP events (roofline.avg_launch_ms) sit above the rocprofv3 trace duration of k_accumulate_seg:
# the same bench command under --kernel-trace with the shipped library and with the side-stream rewrite disabled.
S=${1:-s5}
cd /root/repo
O=/root/repo/gpurun_out/$S
mkdir -p $O
Q="--no-cpu-baseline --no-host-entry --no-pipeline --no-also --no-next-rows"
cd /tmp && export TMPDIR=/tmp
for v in default nofork; do
  if [ $v = default ]; then unset GMSM_LIB; else export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_$v/libgmsm_ab.so; fi
  python /root/repo/bench.py $Q > $O/plain_$v.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t --output-format csv -- python /root/repo/bench.py $Q > $O/traced_$v.json 2> $O/trace_$v.log
  python - <<PY
import json
for tag in ("plain", "traced"):
    d = json.loads(open("$O/%s_$v.json" % tag).read().strip().splitlines()[-1])
    print("$v", tag, "events avg_launch_ms", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), "us; ms_per_step", round(d["ms_per_step"], 4), {k: round(x * 1e3, 1) for k, x in d["stage_ms"].items()})
PY
  python /root/repo/tools/trace_gaps.py $O/trace_$v
  find $O/trace_$v -name "*kernel_trace.csv" -delete; find $O/trace_$v -name "*agent_info.csv" -delete
done 2>&1 | tee $O/events_vs_trace.log
