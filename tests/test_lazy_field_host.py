"""The lazy-limb field code of the kernels (gmsm_fieldu.h, gmsm_field2u.h) compiled for the HOST with g++ and checked
against itself in two formulations (tests/c/lazy_field_check.cpp): double product with one Montgomery reduction vs two
reduced products, lazy-reduction Fp2 product vs Karatsuba, on random values and on every edge of the reduced class
[0, 4q) - the class boundary 4q - 1 is exactly where a 4q-based negation underflows. No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_field_formulations_agree(tmp_path):
    exe = tmp_path / "lazy_field_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__host__=", "-D__device__=", "-D__noinline__=", "-o", str(exe),
                           os.path.join(ROOT, "tests", "c", "lazy_field_check.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("0 mismatches") == 3, r.stdout


def test_lazy_fft_butterflies_match_saturated_field(tmp_path):
    """tests/c/lazy_fft_check.cpp: 2^11-point DIF and DIT transforms on the lazy-limb butterflies (gmsm_fft_lazy.h, eleven
    stages without a canonical reduction = the longest chain a device pass runs) against the saturated field, for the
    three scalar fields, with the value class checked after every butterfly. The saturated field code uses clang's
    carry builtins, so this one is compiled with the ROCm clang++."""
    import pytest
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("ROCm clang++ not found")
    exe = tmp_path / "lazy_fft_check"
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-D__host__=", "-D__device__=", "-D__noinline__=", "-o", str(exe),
                           os.path.join(ROOT, "tests", "c", "lazy_fft_check.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("0 mismatches, 0 class violations") == 3, r.stdout


def test_signed_limb_mixed_addition_matches_unsigned_form(tmp_path):
    """tests/c/lazy_signed_check.cpp: the signed-limb mixed addition of the prime-field accumulation loops (madd_s) against
    the unsigned bound-tracked form (madd_u) on the three base fields: same residues after every step of chains with
    edge-valued coordinates, doublings and cancellations, limbs within what the signed product scans admit, and finished
    records in the class the records in memory use."""
    import pytest
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("ROCm clang++ not found")
    exe = tmp_path / "lazy_signed_check"
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-D__host__=", "-D__device__=", "-D__noinline__=", "-D__forceinline__=inline",
                           "-o", str(exe), os.path.join(ROOT, "tests", "c", "lazy_signed_check.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count(" 0 mismatches") == 5, r.stdout


def test_host_squaring_matches_the_product(tmp_path):
    """tests/c/host_sqr_check.cpp: the dedicated squaring the host-side window fold uses (gmsm_field.h fp_sqr, host branch:
    msmReduceChunk's doublings, multiexp.go:302-315) returns the limbs of fp_mul(x, x) for the six fields, on the edges and on
    random chains."""
    import pytest
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("ROCm clang++ not found")
    exe = tmp_path / "host_sqr_check"
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-D__host__=", "-D__device__=", "-D__noinline__=", "-o", str(exe),
                           os.path.join(ROOT, "tests", "c", "host_sqr_check.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count(": 0 mismatches") == 6, r.stdout
