"""Static checks on the built gfx950 objects (no GPU): the code-generation trap of tools/check_long_branch.py - a long
branch expanded through s[30:31], the return-address registers, inside a device function that is larger than a short
branch reaches - must not be present in anything that ships. Found in round 2: the out-of-line group addition of the
28-word element types (UnsatOpsMid) never returned on the GPU."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_long_branch_through_the_return_address():
    objs = sorted(glob.glob(os.path.join(ROOT, "gnark-crypto_amd", "csrc", "build", "group*.o")))
    if len(objs) < 6 or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("group objects not built here (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_long_branch.py")] + objs, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "long-branch check: ok" in r.stdout, r.stdout + r.stderr
