#!/bin/bash
# the round's last GPU seconds: the full GPU suite on the shipped library (inlined combine / fix-up for the wide groups)
out=/root/repo/gpurun_out/r2r
mkdir -p $out
cd /root/repo
timeout 124 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" | tee -a $out/pytest.log
tail -3 $out/pytest.log
