#!/bin/bash
mkdir -p gpurun_out/r2f
cd /root/repo
(timeout 900 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -30) > gpurun_out/r2f/pytest_ingest.log
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r2f/pytest.log
