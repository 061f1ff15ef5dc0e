#!/bin/bash
# One gpurun call of round 4: the GPU suite, the default bench line, the N > 1 rehearsals at the driver's sizes and the
# fr/fft profile. Everything lands under gpurun_out/$1/.
S=${1:-s1}
cd /root/repo
mkdir -p gpurun_out/$S
O=gpurun_out/$S
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log )
tail -3 $O/gputest.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err )
tail -c 600 $O/bench.json; echo
( timeout 900 python bench.py --gpus 2 --oversubscribe > $O/rehearsal_2.json 2> $O/rehearsal_2.err; echo "rc=$?" >> $O/rehearsal_2.err )
( timeout 1200 python bench.py --gpus 8 --oversubscribe > $O/rehearsal_8.json 2> $O/rehearsal_8.err; echo "rc=$?" >> $O/rehearsal_8.err )
tail -c 700 $O/rehearsal_8.json; echo; tail -2 $O/rehearsal_8.err
tools/profile_fft.sh $S/fft_prof bn254 20 24
( timeout 300 tools/ubench_batch_affine 20 23; timeout 300 tools/ubench_batch_affine 24 23 ) > $O/batch_affine.log 2>&1
cat $O/batch_affine.log
