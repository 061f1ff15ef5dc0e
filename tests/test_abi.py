"""C-ABI checks that need no GPU: the library loads, exports every symbol include/gmsm.h declares, reproduces the
reference's argument errors, fails loudly (no CPU fallback) when no device exists, and its host-side pieces (window
fold, FromJacobian, base generator) agree with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ALL_GROUPS, random_scalars, rng_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def test_library_exports_every_declared_symbol(gm):
    hdr = open(os.path.join(ROOT, "include", "gmsm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gmsm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = gm._lib.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(gm._lib.ABI_SYMBOLS)
    assert lib.gmsm_version().startswith(b"gmsm")


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_layout_sizes(gm, curve, which):
    lib = gm._lib.load()
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    assert lib.gmsm_affine_limbs(g.gid) == g.aff_limbs
    assert lib.gmsm_scalar_limbs(g.gid) == g.fr_limbs
    sizes = {"bn254": (8, 16), "bls12_381": (12, 24), "bw6_761": (24, 24)}[curve]
    assert g.aff_limbs == sizes[0 if which == "g1" else 1]
    # computeNbChunks (multiexp.go:681)
    for c in (2, 5, 11, 16):
        assert lib.gmsm_num_windows(g.gid, c) == (g.curve.fr_bits + c - 1) // c


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_default_window_table(gm, curve, which):
    """gmsm_default_window_bits: the measured per-group table (gmsm_context.h, profiles/r02_window_sweeps.log). It is a
    pure cost choice, but the sharded paths rely on every rank getting the same answer for the same n, on the width
    staying inside what gmsm_window_sums_* accept, and on the top window never being the few-bit kind that serialises
    the sort."""
    gm.set_option("window_bits", 0)
    g = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    fr_bits = g.curve.fr_bits
    prev = None
    for lg in range(0, 31):
        for n in {1 << lg, (1 << lg) + 1, (1 << (lg + 1)) - 1}:
            c = g.default_window_bits(n)
            assert 2 <= c <= 17, (n, c)
            nwin = g.num_windows(c)
            assert nwin == (fr_bits + c - 1) // c
            top_bits = fr_bits - (nwin - 1) * c
            assert top_bits >= 6 or n < (1 << 17), (n, c, top_bits)  # narrow top windows only where they were measured to win
            assert c == g.default_window_bits(1 << lg), "one width per power-of-two band"
        if prev is not None and lg >= 20:  # below 2^19 the measured optimum moves both ways (profiles/r06_glv_width_sweep.log)
            assert c >= prev, "the width does not shrink as n grows (large n)"
        prev = c
    table = {("bn254", "g1"): {20: 16, 21: 17, 24: 17, 26: 17}, ("bn254", "g2"): {20: 16, 22: 17},
             ("bls12_381", "g1"): {22: 16, 24: 17}, ("bls12_381", "g2"): {16: 12, 22: 16},
             ("bw6_761", "g1"): {12: 9, 13: 10, 20: 14, 22: 16}, ("bw6_761", "g2"): {12: 9, 13: 10, 20: 14, 22: 16}}[(curve, which)]
    for lg, c in table.items():
        assert g.default_window_bits(1 << lg) == c, (lg, c)
    with gm.options(window_bits=11):  # gmsm_set_option(GMSM_OPT_WINDOW_BITS): the switch the tests and sweeps use
        assert g.default_window_bits(1 << 20) == 11
    assert g.default_window_bits(1 << 20) == table.get(20, g.default_window_bits(1 << 20))


def test_dump_header_errors_need_no_device(gm, tmp_path):
    """gmsm_bases_register_dump checks the file before it touches the device: missing file, wrong marker
    (utils/unsafe/dump_slice.go:91-103), truncated header, and a length word that promises more points than the file
    holds (ReadSlice would hit io.ErrUnexpectedEOF, dump_slice.go:35-76) all fail with the argument error."""
    g = gm.G1Affine("bn254")
    rb, err = g.register_bases_dump(tmp_path / "missing.dump")
    assert rb is None and "cannot open" in err
    path = tmp_path / "srs.dump"
    path.write_bytes((0xdeadbeef).to_bytes(8, "little")[:5])
    rb, err = g.register_bases_dump(path)
    assert rb is None and "short read (marker)" in err
    path.write_bytes((0xfeedface).to_bytes(8, "little") + (4).to_bytes(8, "little") + bytes(4 * 64))
    rb, err = g.register_bases_dump(path)
    assert rb is None and "marker mismatch" in err
    path.write_bytes((0xdeadbeef).to_bytes(8, "little") + (4).to_bytes(8, "little")[:3])
    rb, err = g.register_bases_dump(path)
    assert rb is None and "short read (length)" in err
    path.write_bytes((0xdeadbeef).to_bytes(8, "little") + (1 << 20).to_bytes(8, "little") + bytes(3 * 64))
    rb, err = g.register_bases_dump(path)
    assert rb is None and "fewer points than its length word says" in err
    rb, err = g.register_bases_dump(path, max_points=4)  # maxElements caps the read, but 4 points are not there either
    assert rb is None and "fewer points than its length word says" in err


def test_reference_argument_errors(gm):
    g = gm.G1Jac("bn254")
    pts = np.zeros((3, 8), dtype=np.uint64)
    _, err = g.MultiExp(pts, np.zeros((2, 4), dtype=np.uint64))
    assert err == "len(points) != len(scalars)"            # ecc/bn254/multiexp.go:63
    _, err = g.MultiExp(pts, np.zeros((3, 4), dtype=np.uint64), gm.MultiExpConfig(NbTasks=1025))
    assert err == "invalid config: config.NbTasks > 1024"  # multiexp.go:70
    lib = gm._lib.load()
    out = np.zeros(12, dtype=np.uint64)
    assert lib.gmsm_multiexp(99, P(pts), 3, P(pts), 3, 0, P(out)) == gm._lib.GMSM_ERR_ARG


def test_no_silent_cpu_fallback(gm):
    """Without a device a compute call must fail with GMSM_ERR_DEVICE, not compute on the host."""
    lib = gm._lib.load()
    if lib.gmsm_device_count() > 0:
        pytest.skip("a GPU is present")
    g = gm.G1Affine("bn254")
    pts = g.generate_points(4, 1, 1)
    res, err = g.MultiExp(pts, np.ones((4, 4), dtype=np.uint64))
    assert res is None and "no usable HIP device" in err and "no CPU fallback" in err


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_host_fold_and_to_affine_match_oracle(gm, oracle_mod, curve, which):
    """gmsm_fold_windows (msmReduceChunk, multiexp.go:302-315) and gmsm_jac_to_affine (FromJacobian, g1.go:150-166) run
    on the host: feed them the oracle's per-window totals and compare with the oracle's own fold."""
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle_mod.Oracle(curve, which)
    rng = rng_for(21, g.gid)
    n = 150
    pts = o.gen_points(n, 17, 5)
    sc = random_scalars(rng, g.curve, n)
    for c in (4, 9, 16):
        digits = o.partition_scalars(sc, c)
        lastc = c + 1 - (o.nb_chunks(c) * c - g.curve.fr_bits)
        totals = np.stack([o.process_chunk(max(c, lastc), pts, digits[j]) for j in range(o.nb_chunks(c))])
        jac = g.fold_windows(totals, c)
        assert (g.jac_to_affine(jac) == o.msm_affine(pts, sc, c=c)).all()
    inf = np.stack([o.xyzz_infinity()] * g.num_windows(16))
    assert (g.jac_to_affine(g.fold_windows(inf, 16)) == 0).all()


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_generate_points_matches_oracle(gm, oracle_mod, curve, which):
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    o = oracle_mod.Oracle(curve, which)
    assert (g.generator == o.generator).all()
    a = g.generate_points(2500, 0xABCDEF, 0x1234567, nthreads=3)
    b = o.gen_points(2500, 0xABCDEF, 0x1234567, nthreads=2)
    assert (a == b).all()


def test_c_client_links_and_fails_loudly_without_a_gpu(gm, tmp_path):
    """The plain-C client of include/gmsm.h (tests/c/abi_client.c, the stand-in for the cgo stub) compiles as C, links
    against libgmsm.so alone and - on a machine without a HIP device - gets GMSM_ERR_DEVICE with the no-fallback message
    instead of a CPU result."""
    import os
    import subprocess
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "gnark-crypto_amd", "csrc")
    exe = str(tmp_path / "abi_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "abi_client.c"), "-o", exe, "-L", libdir, "-lgmsm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the functional run of the client is tests/test_gpu_parity.py::test_c_client_through_the_abi")
    fin = str(tmp_path / "in.bin")
    with open(fin, "wb") as f:
        np.array([2, 8, 4], dtype=np.uint64).tofile(f)
        np.zeros(2 * 8 + 2 * 4, dtype=np.uint64).tofile(f)
    r = subprocess.run([exe, "0", fin, str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr)
    assert "no CPU fallback" in r.stderr


def test_options_and_lifecycle_entries_need_no_device(gm):
    """gmsm_set_option / gmsm_get_option validate and round-trip without touching a device; gmsm_trim and gmsm_shutdown on a
    process that never created a context are no-ops; gmsm_get_devices reports one rank (nothing configured: no spreading)."""
    lib = gm._lib.load()
    assert gm.get_option("window_bits") in range(0, 21) and gm.get_option("tables") in (0, 1, 2)
    with gm.options(window_bits=13, tables=0, max_run=4096, host_ranges=3, fixed_base_bits=11):
        assert [gm.get_option(k) for k in ("window_bits", "tables", "max_run", "host_ranges", "fixed_base_bits")] == [13, 0, 4096, 3, 11]
        assert gm.G1Jac("bn254").default_window_bits(1 << 20) == 13
    assert gm.get_option("max_run") == 0 and gm.get_option("host_ranges") == 0
    # round 5: the fused small-n kernel's switches and the window-group experiment (off by default)
    assert [gm.get_option(k) for k in ("small_bits", "small_max", "split")] == [0, 0, 0]
    with gm.options(small_bits=6, small_max=1024, split=1):
        assert [gm.get_option(k) for k in ("small_bits", "small_max", "split")] == [6, 1024, 1]
    with gm.options(small_bits=1):  # 1 = off
        assert gm.get_option("small_bits") == 1
    assert [gm.get_option(k) for k in ("small_bits", "small_max", "split")] == [0, 0, 0]
    for name, bad in (("window_bits", 1), ("window_bits", 21), ("tables", 3), ("fixed_base_bits", 15), ("small_bits", 8)):
        with pytest.raises(ValueError):
            gm.set_option(name, bad)
    assert lib.gmsm_set_option(99, 1) == gm._lib.GMSM_ERR_ARG and lib.gmsm_get_option(99) == 0
    if lib.gmsm_device_count() == 0:
        assert gm.trim(0) == 0
        gm.shutdown()
        gm.shutdown()
    assert lib.gmsm_get_devices(None, 0) == 1  # NULL output is allowed
    assert len(gm.get_devices()) == 1


def test_race_client_builds_as_c(tmp_path):
    """tests/c/race_client.c (the sanitizer workload) is plain C99 + pthreads against include/gmsm.h and links to the shipped
    library; without a device it says so and exits 77."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "gnark-crypto_amd", "csrc")
    exe = str(tmp_path / "race_client")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "race_client.c"), "-o", exe, "-L", libdir, "-lgmsm", "-lpthread",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_sanitizers.py runs the client")
    r = subprocess.run([exe, "1", "100"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no device" in r.stderr
