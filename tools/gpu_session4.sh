#!/bin/bash
# Final session of round 4: the whole GPU suite, the per-configuration rocprofv3 summaries (kernel trace + FETCH / WRITE
# passes), the bench line, the N > 1 rehearsals at the driver's sizes, the spin-wait A/B, the fuzz and the shard stages.
S=${1:-s4}
cd /root/repo
O=gpurun_out/$S
mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log ); tail -3 $O/gputest.log
cp gpurun_out/sanitizer_*.log $O/ 2>/dev/null
python - > $O/spin_ab.log 2>&1 <<'PY'
import importlib, time, numpy as np, torch
gm = importlib.import_module("gnark-crypto_amd")
g = gm.G1Jac("bn254")
for logn in (16, 20):
    n = 1 << logn
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_p = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_p.data_ptr(), s)
    for rnd in range(3):
        for spin in (0, 4000):
            gm.set_option("spin_wait_us", spin)
            for _ in range(5): g.multiexp_device(d_p.data_ptr(), d_a.data_ptr(), n, s)
            t0 = time.perf_counter()
            for _ in range(40): g.multiexp_device(d_p.data_ptr(), d_a.data_ptr(), n, s)
            print(f"2^{logn} spin_wait_us={spin}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms per call", flush=True)
gm.set_option("spin_wait_us", 4000)
PY
cat $O/spin_ab.log
# quad fix-up (default) against the one-lane fix-up (build_ab_fix1), wide element types, same call
Q="--no-cpu-baseline --no-host-entry --no-pipeline --no-also --no-next-rows"
for cfg in "bw6_761 g1 20" "bls12_381 g2 22" "bn254 g2 20" "bw6_761 g1 16"; do
  set -- $cfg
  for v in default fix1 default fix1; do
    if [ $v = default ]; then unset GMSM_LIB; else export GMSM_LIB=/root/repo/gnark-crypto_amd/csrc/build_ab_$v/libgmsm_ab.so; fi
    timeout 300 python bench.py $Q --curve $1 --group $2 --logn $3 --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $v', round(d['ms_per_step'],3), 'fixup', round(d['stage_ms']['fixup'],4), 'bit_exact?', d.get('bit_exact'))"
  done
done > $O/fixup_ab.log 2>&1
unset GMSM_LIB
cat $O/fixup_ab.log
tools/profile_round.sh $S/prof > $O/profile_round.log 2>&1
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err )
tail -c 900 $O/bench.json; echo
( timeout 900 python bench.py --gpus 2 --oversubscribe > $O/rehearsal_2.json 2> $O/rehearsal_2.err; echo "rc=$?" >> $O/rehearsal_2.err )
( timeout 1200 python bench.py --gpus 8 --oversubscribe > $O/rehearsal_8.json 2> $O/rehearsal_8.err; echo "rc=$?" >> $O/rehearsal_8.err )
tail -c 500 $O/rehearsal_8.json; echo
for grp in "bn254 g1" "bn254 g2" "bls12_381 g1" "bls12_381 g2" "bw6_761 g1" "bw6_761 g2"; do timeout 200 python tools/fuzz_parity.py 20 $grp; done > $O/fuzz.log 2>&1
tail -6 $O/fuzz.log
timeout 300 python tools/shard_stages.py > $O/shard_stages.log 2>&1; cat $O/shard_stages.log
