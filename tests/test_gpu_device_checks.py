"""Device-side unit checks of kernel helpers (tests/hip/*.hip, built by __graft_entry__.build()): small HIP programs that
include the product headers and verify a helper on the GPU against a host model - here the LDS counter updates of the
sort passes with and without run aggregation, from full and partially active waves (gmsm_kernels.h: lds_count)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_lds_counter_helpers_on_the_device():
    exe = os.path.join(ROOT, "tests", "hip", "count_runs_check")
    if not os.path.exists(exe):  # a clean clone that skipped build(): compile here (hipcc is part of the image)
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build_device_checks()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 bad" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
