#!/bin/bash
mkdir -p gpurun_out/r2g
cd /root/repo
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r2g/pytest.log
python bench.py > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err
python bench.py --force-dist --no-cpu-baseline --no-also --steps 10 > gpurun_out/r2g/force_dist.json 2> gpurun_out/r2g/force_dist.err
python bench.py --force-dist --batch 8 --no-cpu-baseline --no-also --steps 5 > gpurun_out/r2g/force_dist_batch.json 2>> gpurun_out/r2g/force_dist.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2g/trace -o bench --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/r2g/trace_bench.json 2> /root/repo/gpurun_out/r2g/trace.log
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/r2g/pmc_fetch -o fetch --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-host-entry --no-pipeline --steps 3 --warmup 1 > /dev/null 2> /root/repo/gpurun_out/r2g/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/r2g/pmc_write -o write --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-host-entry --no-pipeline --steps 3 --warmup 1 > /dev/null 2> /root/repo/gpurun_out/r2g/pmc_write.log
cd /root/repo
ls -la gpurun_out/r2g/* | head -40
du -sh gpurun_out/r2g
