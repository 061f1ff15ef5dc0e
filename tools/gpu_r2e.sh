#!/bin/bash
mkdir -p gpurun_out/r2e
cd /root/repo
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r2e/pytest.log
(time python bench.py) > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench_default.err
for r in 1 2 4; do GMSM_HOST_RANGES=$r python bench.py --no-cpu-baseline --no-pipeline --no-also --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ranges $r 2^20', d['host_entry'])"; done > gpurun_out/r2e/host_ranges.log 2>&1
