for logn in 20 22 24; do
for c in 16 18 20; do
GMSM_C=$c tools/bench_short.sh "logn=$logn c=$c" --logn $logn
done; done
