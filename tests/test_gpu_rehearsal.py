"""Rehearsal of bench.py's N > 1 branch on the 1-GPU box: `bench.py --gpus N --oversubscribe` puts rank r on device
r % device_count and runs the collectives over gloo (RCCL refuses two ranks on one device); everything else - the
self-launch under torch.distributed.run, shard_plan, Exchange, the timed loop with barrier + max over ranks,
sharded_also, host_side_wait, c_abi_sharded over one logical rank per process, the JSON assembly - is the code the
driver's 8-GPU SCALE run executes (ecc/bn254/multiexp.go:148-209 per-window workers, :302-315 fold)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def run_bench(*args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # the contract: ONE JSON line on stdout
    return json.loads(lines[0]), lines[0]


def load_full(tmp_path):
    with open(tmp_path) as f:
        return json.load(f)


@pytest.mark.parametrize("world,extra", [(2, ["--logn", "20", "--also-logn", "21", "--sharded-logns", "19,21,22"]),   # windows, then points
                                         (8, ["--logn", "18", "--also-logn", "20", "--sharded-logns", "19,20", "--shard", "points"])])
def test_bench_sharded_branch_oversubscribed(world, extra, tmp_path):
    sys.path.insert(0, ROOT)
    from tools import bench_line
    full_path = str(tmp_path / "full.json")
    slim, line = run_bench("--gpus", str(world), "--oversubscribe", "--steps", "3", "--warmup", "1", "--full-out", full_path, *extra)
    # the printed line: bounded, schema-clean, and carrying the verdict fields in compact form
    assert len(line.encode()) <= bench_line.MAX_LINE_BYTES and bench_line.validate(line) == []
    assert slim["n_gpus"] == world and slim["backend"] == "gloo" and slim["equal_to_single_gpu_result"] is True
    assert slim["c_abi_sharded"]["equal"] is True and slim["also"][-1]["c_abi_sharded"]["equal"] is True
    assert all(r["bit_exact"] is True and r["n_gpus"] == world and r["exchange_ms"] > 0 and r["stage_ms"]["accumulate"] > 0
               for r in slim["also"])
    assert slim["full_record"] == os.path.relpath(full_path, ROOT)
    rec = load_full(full_path)  # the full record (side file): everything the line summarises
    assert rec["n_gpus"] == world and rec["steps"] == 3 and rec["scaling"] == "strong"
    assert rec["backend"] == "gloo" and rec["rccl_ranks"] == 0 and rec["oversubscribed"] is True
    assert len(rec["devices_seen"]) == world and all(0 <= d < rec["device_count"] for d in rec["devices_seen"])
    assert rec["equal_to_single_gpu_result"] is True
    assert rec["c_abi_sharded"]["equal_to_reference_result"] is True
    assert rec["c_abi_sharded"]["devices"] == rec["devices_seen"]
    also = rec["also"][-1]  # the --also-logn row comes last; the other sharded rows before it
    assert also["n_gpus"] == world and also["bit_exact"] is True
    assert also["c_abi_sharded"]["equal_to_reference_result"] is True
    for row in rec["also"]:  # every sharded row is attributable: per-stage device times (max over ranks) and the exchange
        assert row["bit_exact"] is True and row["n_gpus"] == world
        assert row["stage_ms"]["accumulate"] > 0 and row["exchange_ms"] > 0 and row["compute_ms"] > 0
        assert ("c_abi_sharded" in row) == (row is also)
    assert len(rec["also"]) == len(extra[extra.index("--sharded-logns") + 1].split(","))
    assert rec["exchange_ms"] > 0 and rec["stage_ms"]["accumulate"] > 0
    assert all(v[1] is True for v in rec["n24"]["sharded_rows"].values())
    assert rec["n24"]["bit_exact"] is True and rec["n24"]["c_abi_equal_to_reference_result"] is True
    # the driver keeps the last 2000 characters of the line: the verdict fields must sit there
    kept = line[-2000:]
    for key in ('"n24"', '"equal_to_single_gpu_result"', '"rccl_ranks"', '"devices_seen"'):
        assert key in kept, key
    assert rec["value"] > 0 and rec["ms_per_step"] > 0


def test_bench_refuses_more_ranks_than_devices_without_the_switch():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, capture_output=True,
                       text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "GMSM_BENCH_SHARE_DEVICE"})
    assert p.returncode != 0 and f"needs {n} devices" in p.stderr
