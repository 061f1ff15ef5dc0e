"""The line bench.py prints must stay something the driver parses: round 5's N = 1 line had grown to 26 KB and came back
`parsed: null`. tools/bench_line.py assembles the slim line from the full record; this test runs it on the committed full
records of earlier rounds (canned inputs) and checks size and schema - no GPU, no bench run."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_line  # noqa: E402

CANNED_N1 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_final*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06_bench_full_n1*.json")))
CANNED_N8 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_rehearsal_gpus*_oversubscribed*.json")))


def load(path):
    with open(path) as f:
        text = f.read()
    return json.loads(next(ln for ln in text.splitlines() if ln.startswith("{")))


@pytest.mark.parametrize("path", CANNED_N1, ids=os.path.basename)
def test_slim_line_of_a_full_n1_record_fits_and_carries_the_measurement_blocks(path):
    full = load(path)
    if "full_record" in full:  # already a slim line (a printed line stored under profiles/)
        pytest.skip("stored line, not a full record")
    assert len(json.dumps(full)) > 3 * bench_line.MAX_LINE_BYTES  # the canned record IS the oversized one
    d = bench_line.fit(bench_line.slim_line(full, full_path="bench_full_n1.json"))
    line = bench_line.encode(d)
    assert len(line.encode()) <= bench_line.MAX_LINE_BYTES
    assert "dropped_for_size" not in d  # everything fits without shedding blocks
    assert bench_line.validate(line) == []
    back = json.loads(line)
    for k in ("roofline", "int_roofline", "cpu_baseline", "stage_ms", "n24", "also", "distributions", "small_n", "tail"):
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-3) and back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-3)
    assert back["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert back["roofline"]["traffic"] == int(full["roofline"]["traffic"])
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert len(back["also"]) == len(full["also"])
    for slim_row, row in zip(back["also"], full["also"]):
        assert slim_row["ms"] == pytest.approx(row["ms_per_step"], rel=1e-3) and slim_row["bit_exact"] is row["bit_exact"]
        assert slim_row["roofline"]["frac"] == pytest.approx(row["roofline"]["frac"], rel=1e-3)
    assert back["small_n"]["resident_ms"]["10"] == pytest.approx(next(r for r in full["small_n"]["rows"] if r["logn"] == 10)["resident_ms"], rel=1e-3)


@pytest.mark.parametrize("path", CANNED_N8, ids=os.path.basename)
def test_slim_line_of_a_sharded_record(path):
    full = load(path)
    if "full_record" in full:
        pytest.skip("stored line, not a full record")
    line = bench_line.encode(bench_line.fit(bench_line.slim_line(full)))
    assert len(line.encode()) <= bench_line.MAX_LINE_BYTES and bench_line.validate(line) == []
    back = json.loads(line)
    assert back["n_gpus"] == full["n_gpus"] and back["equal_to_single_gpu_result"] is True
    assert all(r["n_gpus"] == full["n_gpus"] and "stage_ms" in r for r in back["also"])


def test_validator_flags_what_the_driver_would_trip_over():
    full = load(CANNED_N1[0])
    good = bench_line.slim_line(full)
    assert bench_line.validate(good) == []
    bad = dict(good)
    del bad["roofline"]
    assert "roofline missing" in bench_line.validate(bad)
    bad = dict(good, cpu_baseline=None)
    assert "cpu_baseline missing" in bench_line.validate(bad)
    bad = dict(good, vs_baseline=1.0)
    assert any("vs_baseline" in p for p in bench_line.validate(bad))
    bad = dict(good, padding="x" * bench_line.MAX_LINE_BYTES)
    assert any("bytes >" in p for p in bench_line.validate(bad))
    fitted = bench_line.fit(dict(good, padding2="y" * 3000))  # sheds optional blocks, never the measurement blocks
    assert "roofline" in fitted and "cpu_baseline" in fitted and "int_roofline" in fitted
    assert bench_line.validate(bench_line.encode(full)) != []  # the round-5 line itself: too long
