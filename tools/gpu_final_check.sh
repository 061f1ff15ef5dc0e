#!/bin/bash
# The round's closing check on the final build: whole GPU suite, smoke(), the bench line.
cd /root/repo; O=gpurun_out/${1:-final}; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); tail -c 700 $O/bench.json; echo
