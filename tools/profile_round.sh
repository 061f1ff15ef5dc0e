#!/bin/bash
# Profiles of one round on the GPU box: kernel traces and PMC traffic per BASELINE configuration -> gpurun_out/$1/
out=/root/repo/gpurun_out/${1:-prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-entry --no-pipeline --no-also --no-next-rows --full-out /tmp/bench_full_profile.json"  # the profiled commands must not overwrite the round's bench_full_n1.json
run() {  # label, bench args...
  label=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/$label/trace -o t --output-format csv -- python /root/repo/bench.py $Q "$@" > $out/$label/bench.json 2> $out/$label/trace.log
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/$label/fetch -o f --output-format csv -- python /root/repo/bench.py $Q --steps 3 --warmup 1 "$@" > /dev/null 2> $out/$label/fetch.log
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/$label/write -o w --output-format csv -- python /root/repo/bench.py $Q --steps 3 --warmup 1 "$@" > /dev/null 2> $out/$label/write.log
  find $out/$label -name "*kernel_trace.csv" -delete; find $out/$label -name "*agent_info.csv" -delete
}
mkdir -p $out/bn254_g1_20 $out/bn254_g1_22 $out/bn254_g1_24 $out/bn254_g1_26 $out/bn254_g2_20 $out/bls12_381_g1_22 $out/bls12_381_g2_22 $out/bw6_761_g1_20
run bn254_g1_20
run bn254_g1_22 --logn 22 --steps 5
run bn254_g1_24 --logn 24 --steps 5
run bn254_g1_26 --logn 26 --steps 3 --warmup 1
run bn254_g2_20 --curve bn254 --group g2 --logn 20 --steps 5
run bls12_381_g1_22 --curve bls12_381 --group g1 --logn 22 --steps 5
run bls12_381_g2_22 --curve bls12_381 --group g2 --logn 22 --steps 3
run bw6_761_g1_20 --curve bw6_761 --group g1 --logn 20 --steps 3
du -sh $out
cd /root/repo
timeout 60 tools/ubench_ldsagg > $out/ubench_ldsagg.log 2>&1
timeout 120 tools/ubench_fpmul > $out/ubench_fpmul.log 2>&1
timeout 120 tools/ubench_madd > $out/ubench_madd.log 2>&1
timeout 120 tools/ubench_madd_bw6 > $out/ubench_madd_bw6.log 2>&1
timeout 300 python tools/bench_fft.py bn254 16 20 22 24 > $out/fft_bn254.log 2>&1
python tools/bench_fft.py bls12_381 20 24 >> $out/fft_bn254.log 2>&1
python tools/bench_fft.py bw6_761 20 >> $out/fft_bn254.log 2>&1
