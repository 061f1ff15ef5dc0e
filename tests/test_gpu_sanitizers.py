"""ThreadSanitizer / AddressSanitizer runs of the host shim (SURVEY.md §5 "ASan for the host shim"; the reference's CI runs
`go test -race ./ecc/bn254/...`, .github/workflows/pr.yml:63).  The sanitizer libraries are separate builds
(`make -C gnark-crypto_amd/csrc tsan asan`, ~4 minutes each; build() makes them when GMSM_BUILD_SANITIZERS=1): the tests
skip when they have not been built - unless GMSM_REQUIRE_SANITIZERS=1 (tools/gpu_session.sh sets it for the round's suite
runs), which turns the skip into a failure so that a clean clone cannot report green without them unnoticed.  tests/c/race_client.c - six threads over every kind of entry, handles released under use, tables appearing under
use, first-use coset tables, trim under load, shutdown and restart - must finish with equal results and without a
sanitizer report that names libgmsm code."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gnark-crypto_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

pytestmark = pytest.mark.gpu


def clang_rt_dir():
    """directory of libclang_rt.{tsan,asan}-x86_64.so (clang's --print-runtime-dir names a per-target directory that this
    ROCm build does not populate; the shared runtimes sit in lib/linux)"""
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so")
    return os.path.dirname(hits[0]) if hits else None


TEARDOWN_CHECK = 'CHECK failed: sanitizer_allocator_device.h'


@pytest.mark.parametrize("kind,flag", [("tsan", "thread"), ("asan", "address")])
def test_race_client_under_sanitizer(kind, flag, tmp_path):
    lib = os.path.join(CSRC, f"build_{kind}", f"libgmsm_{kind}.so")
    if not os.path.exists(lib):
        if os.environ.get("GMSM_REQUIRE_SANITIZERS") == "1":
            pytest.fail(f"{lib} not built and GMSM_REQUIRE_SANITIZERS=1 (GMSM_BUILD_SANITIZERS=1 python -c 'import __graft_entry__ as g; g.build()')")
        pytest.skip(f"{lib} not built (make -C gnark-crypto_amd/csrc {kind})")
    rt = clang_rt_dir()
    exe = str(tmp_path / f"race_client_{kind}")
    cmd = [CLANG, "-O1", "-g", f"-fsanitize={flag}", "-shared-libsan", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "race_client.c"), "-o", exe, lib, "-lpthread",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"] + (["-Wl,-rpath," + rt] if rt else [])
    subprocess.run(cmd, check=True)
    env = dict(os.environ)
    # reports are about OUR code: the HIP / HSA runtimes are not instrumented (their internal synchronisation is invisible
    # to TSan), so reports whose stacks never enter libgmsm are suppressed by library name
    supp = tmp_path / "supp.txt"
    supp.write_text("called_from_lib:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\nrace:libamdhip64.so\nrace:libhsa-runtime64.so\n"
                    "deadlock:libamdhip64.so\nmutex:libamdhip64.so\n")
    env["TSAN_OPTIONS"] = f"suppressions={supp} halt_on_error=0 exitcode=66 report_signal_unsafe=0"
    env["ASAN_OPTIONS"] = "detect_leaks=0 protect_shadow_gap=0 exitcode=66"  # the HIP runtime keeps its allocations; shadow gap: ROCm maps there
    r = subprocess.run([exe, "4", "20000"], env=env, capture_output=True, text=True, timeout=900)
    log = os.path.join(ROOT, "gpurun_out", f"sanitizer_{kind}.log")
    try:
        os.makedirs(os.path.dirname(log), exist_ok=True)
        with open(log, "w") as f:
            f.write(f"$ {' '.join(cmd)}\n$ race_client 4 20000 -> exit {r.returncode}\n--- stdout\n{r.stdout}\n--- stderr\n{r.stderr[-20000:]}\n")
    except OSError:
        pass
    assert "failure(s)" in r.stdout, (r.returncode, r.stderr[-3000:])
    assert " 0 failure(s)" in r.stdout, r.stdout
    assert "gmsm_" not in "".join(ln for ln in r.stderr.splitlines(True) if ln.lstrip().startswith("#")), r.stderr[-6000:]
    if r.returncode != 0 and TEARDOWN_CHECK in r.stderr and "ERROR: AddressSanitizer" not in r.stderr:
        # Seen once in five runs (profiles/r04_sanitizer_asan_teardown.log): after main() has returned with the client's
        # verdict printed, a thread of the HIP runtime is destroyed after ROCm's ASan runtime has marked its device
        # allocator unloaded, and the runtime's own consistency CHECK aborts the exit. No frame of ours, no memory
        # report: a teardown-order defect between the two runtimes, not a finding about libgmsm.
        return
    assert r.returncode == 0, (r.returncode, r.stderr[-6000:])
