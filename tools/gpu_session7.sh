#!/bin/bash
# Signed-limb mixed addition: the shipped build (madd_s, accumulators stored raw, readers finish them) against
#   build_ab_flushloop (-DGMSM_ACC_RAW_RECORDS=0: the flush converts inside the loop) and
#   build_ab_unsigned  (-DGMSM_SIGNED_MADD=0: madd_u, the round-3 form), each on the four prime-field groups.
# GPU suite on the shipped build first, then the bench line of the three builds, alternated twice.
S=${1:-s7}
cd /root/repo
O=gpurun_out/$S; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
D=/root/repo/gnark-crypto_amd/csrc
for rep in 1 2; do
  ( timeout 600 python bench.py --no-next-rows > $O/bench_signed_$rep.json 2> $O/bench_signed_$rep.err )
  ( GMSM_LIB=$D/build_ab_flushloop/libgmsm_ab.so timeout 600 python bench.py --no-next-rows > $O/bench_flushloop_$rep.json 2> $O/bench_flushloop_$rep.err )
  ( GMSM_LIB=$D/build_ab_unsigned/libgmsm_ab.so timeout 600 python bench.py --no-next-rows > $O/bench_unsigned_$rep.json 2> $O/bench_unsigned_$rep.err )
done
python tools/signed_ab_report.py $O | tee $O/signed_ab.log
