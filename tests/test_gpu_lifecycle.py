"""Library lifecycle (round 4): gmsm_trim gives the grow-only scratch of idle workspaces back to the device, gmsm_shutdown
releases everything and leaves the library usable.  The reference's buffers are per call and garbage-collected
(ecc/bn254/multiexp.go:148-176); a long-lived Go process needs the equivalent."""
import numpy as np
import pytest

from conftest import random_scalars, rng_for

pytestmark = pytest.mark.gpu


def free_bytes(torch):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def test_trim_returns_the_scratch_of_a_large_call(gm, oracle_mod):
    """A 2^24-point MultiExp grows one workspace to several GB; after gmsm_trim(0) the device is back to within 1 GiB of
    where it started, and the next call simply grows the scratch again (same result)."""
    import torch
    g = gm.G1Jac("bn254")
    n = 1 << 24
    small = 1 << 12
    o = oracle_mod.Oracle("bn254", "g1")
    pts_s = o.gen_points(small, 41, 7, nthreads=4)
    sc_s = random_scalars(rng_for(91), g.curve, small)
    jac, err = g.MultiExp(pts_s, sc_s)  # contexts, streams, LDS attributes exist from here on
    assert err is None
    expected_small = g.jac_to_affine(jac)
    assert (expected_small == o.msm_affine(pts_s, sc_s, nthreads=4)).all()
    gm.trim(0)
    start = free_bytes(torch)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**62, size=(n, g.fr_limbs), dtype=np.uint64)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    g.batch_scalar_mul_device(g.generator, d_a.data_ptr(), n, d_pts.data_ptr(), stream)
    j1 = g.multiexp_device(d_pts.data_ptr(), d_a.data_ptr(), n, stream)
    held = free_bytes(torch)
    del d_pts, d_a
    inputs = (64 + 32) * n
    assert start - held > inputs + (2 << 30), "a 2^24 call should hold GBs of scratch"
    freed = gm.trim(0)
    assert freed > (2 << 30)
    after = free_bytes(torch)
    assert start - after < (1 << 30), (start, after)
    # keep_bytes: small buffers survive a partial trim
    jac, err = g.MultiExp(pts_s, sc_s)
    assert err is None and (g.jac_to_affine(jac) == expected_small).all()
    assert gm.trim(1 << 30) == 0
    assert gm.trim(0) > 0
    assert j1 is not None


def test_shutdown_releases_handles_and_the_library_comes_back(gm, oracle_mod):
    import torch
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = 5000
    pts = o.gen_points(n, 43, 9, nthreads=4)
    sc = random_scalars(rng_for(92), g.curve, n)
    expected = o.msm_affine(pts, sc, nthreads=4)
    rb = g.register_bases(points=pts)
    d = gm.fft.NewDomain("bn254", 1 << 10)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    t = rb.submit(d_sc.data_ptr(), n)
    with pytest.raises(RuntimeError, match="has not been collected"):
        gm.shutdown()
    assert (g.jac_to_affine(rb.collect(t)) == expected).all()
    gm.shutdown()
    jac, err = rb.MultiExp(sc)  # the handle went with the shutdown
    assert jac is None and "unknown bases handle" in err
    with pytest.raises(ValueError, match="unknown fft domain handle"):
        d.FFT(np.zeros((1 << 10, 4), dtype=np.uint64), gm.fft.DIF)
    rb.handle = 0
    d.handle = 0
    # ... and everything works again: contexts, workspaces and handles reappear on first use, old handle values stay dead
    jac, err = g.MultiExp(pts, sc)
    assert err is None and (g.jac_to_affine(jac) == expected).all()
    rb2 = g.register_bases(points=pts)
    try:
        assert (g.jac_to_affine(rb2.multiexp_device(d_sc.data_ptr(), n)) == expected).all()
    finally:
        rb2.release()
