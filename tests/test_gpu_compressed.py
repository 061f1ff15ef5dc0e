"""The Encoder's default (compressed) point format on the device (gmsm_decompress.h; gmsm_points_from_compressed,
gmsm_points_compress, gmsm_bases_register_compressed) against the big-integer restatement of (*G1Affine).Bytes / setBytes in
tests/compressed_points.py (ecc/bn254/marshal.go:801-823, :862-948 and the G2 / 3-bit-flag twins): every group, both flags,
infinity, every error the reference words, the subgroup step, and a round trip at a size no Python model reaches."""
import numpy as np
import pytest

import compressed_points as cp
from conftest import ALL_GROUPS
from subgroup_points import curve_b, curve_points, sqrt_fp, sqrt_fp2, times_r

pytestmark = pytest.mark.gpu


def fixture(gm, pyref_mod, curve, which):
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    return g, pyref_mod.Group(g.curve, which)


def limbs(pg, P):
    return np.array(pg.point_to_limbs(P), dtype=np.uint64)


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_decode_and_encode_match_the_model(gm, pyref_mod, curve, which):
    g, pg = fixture(gm, pyref_mod, curve, which)
    n = 48
    pts = g.generate_points(n, 0xC0DEC, 0x51)
    pts[7] = 0                                              # infinity
    for k in (3, 11, 12, 30):                               # make sure both halves occur: negate some
        P = pg.point_from_limbs(pts[k])
        pts[k] = limbs(pg, (P[0], cp.neg(pg, P[1])))
    model = b"".join(cp.encode_compressed(pg, pg.point_from_limbs(pts[i])) for i in range(n))
    assert len(model) == n * g.compressed_point_bytes
    bits, small, large, _ = cp.flags(curve)
    seen = {model[i * g.compressed_point_bytes] >> (8 - bits) for i in range(n)}
    assert {small, large} <= seen
    got, err = g.DecodeCompressed(model)
    assert err is None and (got == pts).all()
    enc, err = g.Compress(points=pts)
    assert err is None and bytes(enc) == model
    # device pointer in, device pointer out
    import torch
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    enc2, err = g.Compress(d_points=d_pts.data_ptr(), n=n)
    assert err is None and bytes(enc2) == model
    d_out = torch.zeros_like(d_pts)
    m, err = g.DecodeCompressed(model, d_out=d_out.data_ptr())
    assert err is None and m == n and (d_out.cpu().numpy().view(np.uint64) == pts).all()
    # the slice form of the Decoder: length prefix, encoding told by the first point's flag
    got, err = g.DecodeSlice(n.to_bytes(4, "big") + model)
    assert err is None and (got == pts).all()
    got, err = g.DecodeSlice(n.to_bytes(4, "big") + model[:-1])
    assert got is None and err == "short buffer"


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_decode_rejects_what_the_reference_rejects(gm, pyref_mod, curve, which):
    g, pg = fixture(gm, pyref_mod, curve, which)
    bits, small, large, inf = cp.flags(curve)
    size = g.compressed_point_bytes
    nb = 8 * g.curve.fp_limbs
    n = 10
    pts = g.generate_points(n, 31, 17)
    good = [cp.encode_compressed(pg, pg.point_from_limbs(pts[i])) for i in range(n)]

    def decode(mods, **kw):
        recs = list(good)
        for i, b in mods.items():
            recs[i] = bytes(b)
        return g.DecodeCompressed(b"".join(recs), **kw)

    # infinity flag with a non-zero payload (ErrInvalidInfinityEncoding) - in the last byte and in the first
    for pos in (size - 1, 0, 1):
        bad = bytearray(cp.encode_compressed(pg, None))
        bad[pos] |= 1
        out, err = decode({4: bad})
        assert out is None and "point 4" in err and cp.ERR_INFINITY in err, (pos, err)
    # uncompressed and undefined flags
    undefined = [0] + ([] if bits == 2 else [0b001, 0b010, 0b011, 0b111])
    for f in undefined:
        bad = bytearray(good[6])
        bad[0] = (bad[0] & (0xff >> bits)) | (f << (8 - bits))
        out, err = decode({6: bad})
        assert out is None and "point 6" in err and cp.ERR_FLAG in err, (f, err)
    # X not below the modulus: q itself, and q + 1 in the low coordinate of an Fp2 point
    bad = bytearray(g.curve.p.to_bytes(nb, "big") * pg.ext)
    bad[0] |= small << (8 - bits)
    out, err = decode({2: bad})
    assert out is None and "point 2" in err and cp.ERR_ELEMENT in err
    if pg.ext == 2:
        bad = bytearray((5).to_bytes(nb, "big") + (g.curve.p + 1).to_bytes(nb, "big"))
        bad[0] |= large << (8 - bits)
        out, err = decode({2: bad})
        assert out is None and "point 2" in err and cp.ERR_ELEMENT in err
    # an X with no Y: the model finds one, the device words the same error
    t = 2
    while True:
        if pg.ext == 1:
            x, has = t, sqrt_fp((t * t * t + curve_b(pyref_mod, pg)) % pg.p, pg.p) is not None
            enc = bytearray(x.to_bytes(nb, "big"))
        else:
            x = pyref_mod.Fp2(t, 3, pg.p)
            has = sqrt_fp2(pyref_mod, x * x * x + curve_b(pyref_mod, pg)) is not None
            enc = bytearray(x.a1.to_bytes(nb, "big") + x.a0.to_bytes(nb, "big"))
        if not has:
            break
        t += 1
    enc[0] |= large << (8 - bits)
    with pytest.raises(ValueError, match="square root"):
        cp.decode_compressed(pyref_mod, pg, bytes(enc))
    out, err = decode({8: enc}, subgroup_check=False)
    assert out is None and "point 8" in err and cp.ERR_SQRT in err
    # two offenders: the first one is reported
    out, err = decode({8: enc, 5: bad if pg.ext == 1 else bytearray(enc)})
    assert out is None and "point 5" in err


@pytest.mark.parametrize("curve,which", [gw for gw in ALL_GROUPS if gw != ("bn254", "g1")])
def test_subgroup_step(gm, pyref_mod, curve, which):
    """A compressed curve point outside the r-torsion decodes (NoSubgroupChecks, marshal.go:426) and is refused by default
    with the reference's text; decoding itself is the same either way."""
    g, pg = fixture(gm, pyref_mod, curve, which)
    n = 9
    pts = g.generate_points(n, 5, 3)
    recs = [cp.encode_compressed(pg, pg.point_from_limbs(pts[i])) for i in range(n)]
    outside = next(P for P in curve_points(pyref_mod, pg, 200, start=4) if times_r(pg, P) is not None)
    recs[4] = cp.encode_compressed(pg, outside)
    out, err = g.DecodeCompressed(b"".join(recs))
    assert out is None and "point 4" in err and "subgroup check failed" in err
    out, err = g.DecodeCompressed(b"".join(recs), subgroup_check=False)
    assert err is None and pg.point_from_limbs(out[4]) == outside
    assert pg.point_from_limbs(out[4]) == cp.decode_compressed(pyref_mod, pg, recs[4])
    # an undecodable point before it wins, one after it loses
    bits, small, large, inf = cp.flags(curve)
    bad = bytearray(cp.encode_compressed(pg, None))
    bad[-1] = 7
    for pos, expect in ((2, "point 2"), (6, "point 4")):
        r2 = list(recs)
        r2[pos] = bytes(bad)
        out, err = g.DecodeCompressed(b"".join(r2))
        assert out is None and expect in err, (pos, err)


@pytest.mark.parametrize("curve,which", [("bn254", "g2"), ("bls12_381", "g2")])
def test_fp2_square_root_cases(gm, pyref_mod, curve, which):
    """The device's Fp2 root (norm, then two roots in Fp: gmsm_decompress.h) on what the decoder rarely meets: operands in Fp
    (a1 = 0: residues and non-residues of Fp - the latter have the root sqrt(-a0) u), purely imaginary operands, random
    squares and non-squares; checked by squaring with big integers, existence against the reference's algorithm."""
    g, pg = fixture(gm, pyref_mod, curve, which)
    p = pg.p
    rng = np.random.default_rng(99)
    rnd = lambda: int.from_bytes(rng.bytes(64), "big") % p
    vals = []
    for _ in range(12):
        vals.append(pyref_mod.Fp2(rnd(), 0, p))                  # in Fp (half of them non-residues of Fp)
        vals.append(pyref_mod.Fp2(0, rnd(), p))                  # purely imaginary
        z = pyref_mod.Fp2(rnd(), rnd(), p)
        vals.append(z * z)                                       # a square
        vals.append(z)                                           # whatever
    vals = [v for v in vals if not v.is_zero()]
    n = g.curve.fp_limbs
    a = np.array([pyref_mod.fp_to_mont(g.curve, v.a0) + pyref_mod.fp_to_mont(g.curve, v.a1) for v in vals], dtype=np.uint64)
    out = np.zeros_like(a)
    L = gm._lib.load()
    P = lambda x: x.ctypes.data_as(gm._lib.ctypes.POINTER(gm._lib.ctypes.c_uint64))
    assert L.gmsm_debug_field_op(g.gid, 3, 7, P(a), None, len(vals), P(out)) == 0, gm._lib.last_error()
    roots = 0
    for v, row in zip(vals, out):
        r = pyref_mod.Fp2(pyref_mod.fp_from_mont(g.curve, [int(x) for x in row[:n]]),
                          pyref_mod.fp_from_mont(g.curve, [int(x) for x in row[n:]]), p)
        exists = sqrt_fp2(pyref_mod, v) is not None
        assert (not r.is_zero()) == exists, v
        if exists:
            assert r * r == v, v
            roots += 1
    assert 24 <= roots < len(vals)   # every Fp and every square has one; some of the rest do not


@pytest.mark.parametrize("curve,which,logn", [("bn254", "g1", 18), ("bls12_381", "g1", 16), ("bls12_381", "g2", 15), ("bw6_761", "g1", 14)])
def test_round_trip_at_size(gm, curve, which, logn):
    """decompress(compress(P)) == P for 2^14..2^18 device-generated points (the size-independent property; the model above pins
    what the bytes mean), then the same bytes registered as bases give the MultiExp of the points themselves."""
    import torch
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    gj = (gm.G1Jac if which == "g1" else gm.G2Jac)(curve)
    n = 1 << logn
    rng = np.random.default_rng([7, logn])
    from conftest import random_scalars
    sc = random_scalars(rng, g.curve, n)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    d_pts = torch.empty((n, g.aff_limbs), dtype=torch.int64, device="cuda")
    gj.batch_scalar_mul_device(gj.generator, d_sc.data_ptr(), n, d_pts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    pts = d_pts.cpu().numpy().view(np.uint64)
    pts[[3, n - 1]] = 0
    comp, err = g.Compress(points=pts)
    assert err is None and comp.size == n * g.compressed_point_bytes
    back, err = g.DecodeCompressed(comp)
    assert err is None and (back == pts).all()
    rb, err = g.register_bases_compressed(comp)
    assert err is None
    try:
        m = min(n, 1 << 14)
        jac, err = rb.MultiExp(sc[:m])
        want, err2 = g.MultiExp(pts[:m], sc[:m])
        assert err is None and err2 is None and (g.jac_to_affine(jac) == want).all()
    finally:
        rb.release()
