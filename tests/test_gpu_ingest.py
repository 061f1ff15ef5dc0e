"""N4: point ingest on the device - the reference's uncompressed wire format with the Decoder's
checks (ecc/bn254/marshal.go:826, :862-905, :250-275; 3-flag-bit variant ecc/bls12-381/marshal.go:27-34), validation of
limb-form points, and SRS dumps (utils/unsafe/dump_slice.go, ecc/bn254/kzg/marshal.go:65-113) straight into HBM.

The expected values come from the independent big-int model (oracle/pyref.py): encoding by the reference's rules,
curve membership, and [r]P for the subgroup decision."""
import os
import struct

import numpy as np
import pytest

from conftest import ALL_GROUPS, random_scalars, rng_for

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- the reference's encoding, restated with big ints
def flag_bits(curve_name):
    return 2 if curve_name == "bn254" else 3  # marshal.go:25-31 / bls12-381/marshal.go:27-34


def coord_values(pg, P):
    """Wire order of the base-field values of an affine point: X | Y, Fp2 coordinates as A1 | A0 (marshal.go:1090-1104)."""
    x, y = P
    if pg.ext == 1:
        return [x, y]
    return [x.a1, x.a0, y.a1, y.a0]


def encode_raw(pg, P):
    """(*G1Affine).RawBytes (marshal.go:826-846): big-endian regular form, flag bits in the first byte."""
    nb = 8 * pg.c.fp_limbs
    size = 2 * pg.ext * nb
    if P is None:
        out = bytearray(size)
        if pg.c.name != "bn254":
            out[0] = 0b010 << 5  # mUncompressedInfinity
        return bytes(out)
    return b"".join(v.to_bytes(nb, "big") for v in coord_values(pg, P))  # mUncompressed = 0: nothing to OR in


from subgroup_points import curve_b, curve_point_outside_subgroup, curve_points, sqrt_fp, sqrt_fp2, times_r  # noqa: E402,F401


def group_fixture(gm, pyref_mod, curve, which):
    g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
    pg = pyref_mod.Group(g.curve, which)
    return g, pg


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_decode_raw_round_trip(gm, pyref_mod, curve, which):
    g, pg = group_fixture(gm, pyref_mod, curve, which)
    n = 40
    pts = g.generate_points(n, 0xABCDEF, 0x1234567)
    pts[5] = 0  # infinity
    raw = b"".join(encode_raw(pg, pg.point_from_limbs(pts[i])) for i in range(n))
    assert len(raw) == n * g.raw_point_bytes
    got, err = g.DecodeRaw(raw)
    assert err is None
    assert (got == pts).all()
    # the slice form of the Decoder: uint32 big-endian length first
    got2, err2 = g.DecodeSlice(struct.pack(">I", n) + raw)
    assert err2 is None and (got2 == pts).all()
    assert g.DecodeSlice(struct.pack(">I", n + 1) + raw) == (None, "short buffer")


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_decode_raw_rejects_what_the_reference_rejects(gm, pyref_mod, curve, which):
    g, pg = group_fixture(gm, pyref_mod, curve, which)
    n = 12
    pts = g.generate_points(n, 99, 7)
    good = [encode_raw(pg, pg.point_from_limbs(pts[i])) for i in range(n)]
    size = g.raw_point_bytes
    nb = 8 * g.curve.fp_limbs

    def decode(mods, **kw):
        recs = list(good)
        for i, rec in mods.items():
            recs[i] = rec
        return g.DecodeRaw(b"".join(recs), **kw)

    # off the curve: y + 1
    P = pg.point_from_limbs(pts[3])
    y1 = (P[1] + 1) % pg.p if pg.ext == 1 else P[1] + pyref_mod.Fp2(1, 0, pg.p)
    out, err = decode({3: encode_raw(pg, (P[0], y1))})
    assert out is None and "point 3" in err and "not on the curve" in err
    # ... which passes when no check is requested (setBytes(buf, false) checks nothing, marshal.go:894-905)
    out, err = decode({3: encode_raw(pg, (P[0], y1))}, subgroup_check=False, on_curve_check=False)
    assert err is None and pg.point_from_limbs(out[3]) == (P[0], y1)
    # first offender wins
    out, err = decode({7: encode_raw(pg, (P[0], y1)), 2: encode_raw(pg, (P[0], y1))})
    assert "point 2" in err
    # a coordinate that is not below the modulus (SetBytesCanonical)
    bad = bytearray(good[4])
    bad[size - nb:size] = g.curve.p.to_bytes(nb, "big")
    out, err = decode({4: bytes(bad)})
    assert out is None and "point 4" in err and "fp.Element" in err
    # compressed / undefined flag bits are not the raw format
    shift = 8 - flag_bits(curve)
    for flag in ([0b10, 0b11, 0b01] if curve == "bn254" else [0b100, 0b101, 0b110, 0b001, 0b011, 0b111]):
        bad = bytearray(good[1])
        bad[0] |= flag << shift
        out, err = decode({1: bytes(bad)})
        assert out is None and "point 1" in err and "encoding" in err, (flag, err)
    # infinity flag with a non-zero payload
    if curve != "bn254":
        bad = bytearray(size)
        bad[0] = 0b010 << 5
        bad[size - 1] = 1
        out, err = decode({6: bytes(bad)})
        assert out is None and "point 6" in err and "infinity" in err
    # a well-formed infinity is accepted and decodes to (0, 0)
    out, err = decode({6: encode_raw(pg, None)})
    assert err is None and not out[6].any()


@pytest.mark.parametrize("curve,which", [gw for gw in ALL_GROUPS if gw != ("bn254", "g1")])
def test_subgroup_check(gm, pyref_mod, curve, which):
    """Cofactor != 1: a curve point outside the r-torsion is refused by the default Decoder and accepted with
    NoSubgroupChecks (marshal.go:426); the decision equals [r]P == infinity from the big-int model."""
    g, pg = group_fixture(gm, pyref_mod, curve, which)
    n = 9
    pts = g.generate_points(n, 5, 3)
    recs = [encode_raw(pg, pg.point_from_limbs(pts[i])) for i in range(n)]
    outside = curve_point_outside_subgroup(pyref_mod, pg)
    recs[4] = encode_raw(pg, outside)
    out, err = g.DecodeRaw(b"".join(recs))
    assert out is None and "point 4" in err and "subgroup check failed" in err
    out, err = g.DecodeRaw(b"".join(recs), subgroup_check=False)
    assert err is None and pg.point_from_limbs(out[4]) == outside
    # the same through the limb-form validation entry
    ok, err = g.ValidatePoints(points=out)
    assert not ok and "point 4" in err
    ok, err = g.ValidatePoints(points=out, subgroup_check=False)
    assert ok and err is None
    ok, err = g.ValidatePoints(points=pts)
    assert ok


@pytest.mark.parametrize("curve,which", [gw for gw in ALL_GROUPS if gw != ("bn254", "g1")])
def test_subgroup_identity_against_the_definition(gm, pyref_mod, curve, which):
    """Level 2 (the reference's endomorphism identity, gmsm_subgroup.h) and level 3 ([r]P = infinity) decide the same on
    every kind of curve point: r-torsion points, points with a cofactor component, points whose order divides the cofactor
    (the ones a shortcut is most likely to let through), sums of both kinds, infinity - point by point."""
    g, pg = group_fixture(gm, pyref_mod, curve, which)
    good = [pg.point_from_limbs(p) for p in g.generate_points(6, 11, 7)]
    outside = curve_points(pyref_mod, pg, 5, start=3)
    cof = [times_r(pg, P) for P in outside]                       # order divides the cofactor
    mixed = [pg.add(T, good[i]) for i, T in enumerate(cof)]      # r-torsion + cofactor-torsion
    cases = [(P, True) for P in good] + [(None, True)] + [(P, times_r(pg, P) is None) for P in outside + cof + mixed]
    assert sum(not ok for _, ok in cases) >= 12
    for P, expect in cases:
        limbs = np.array(pg.point_to_limbs(P), dtype=np.uint64)
        assert pyref_mod.is_in_subgroup_endo(pg, P) == expect
        for by_def in (False, True):
            ok, err = g.ValidatePoints(points=limbs[None, :], by_definition=by_def)
            assert ok == expect, (P, by_def, err)
            assert ok or "subgroup check failed" in err
    # a vector: the first offender is reported, by both levels
    vec = np.array([pg.point_to_limbs(P) for P in good[:3] + [mixed[1]] + good[3:] + [cof[0]]], dtype=np.uint64)
    for by_def in (False, True):
        ok, err = g.ValidatePoints(points=vec, by_definition=by_def)
        assert not ok and "point 3" in err


def test_bn254_g1_has_no_torsion_check(gm, pyref_mod):
    """Prime-order curve: IsInSubGroup is IsOnCurve (g1.go:475-482)."""
    g, pg = group_fixture(gm, pyref_mod, "bn254", "g1")
    x = 1
    while sqrt_fp((x * x * x + 3) % pg.p, pg.p) is None:
        x += 1
    P = (x, sqrt_fp((x * x * x + 3) % pg.p, pg.p))
    assert times_r(pg, P) is None
    out, err = g.DecodeRaw(encode_raw(pg, P))
    assert err is None and pg.point_from_limbs(out[0]) == P


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bls12_381", "g2")])
def test_register_raw_then_multiexp(gm, oracle_mod, pyref_mod, curve, which):
    g, pg = group_fixture(gm, pyref_mod, curve, which)
    o = oracle_mod.Oracle(curve, which)
    n = 300
    pts = g.generate_points(n, 1234, 77)
    pts[17] = 0
    raw = b"".join(encode_raw(pg, pg.point_from_limbs(pts[i])) for i in range(n))
    rb, err = g.register_bases_raw(raw)
    assert err is None and rb.n == n
    sc = random_scalars(rng_for(4, n), g.curve, n)
    jac, err = rb.MultiExp(sc)
    assert err is None
    assert (g.jac_to_affine(jac) == o.msm_affine(pts, sc)).all()
    jac, err = rb.MultiExp(sc[:100])  # prefix, like kzg.Commit over pk.G1[:len(p)]
    assert (g.jac_to_affine(jac) == o.msm_affine(pts[:100], sc[:100])).all()
    rb.release()
    bad = bytearray(raw)
    bad[5 * g.raw_point_bytes + g.raw_point_bytes - 1] ^= 1
    rb, err = g.register_bases_raw(bytes(bad))
    assert rb is None and "point 5" in err


def test_register_dump(gm, oracle_mod, tmp_path):
    """A file shaped like kzg.SRS.WriteDump's output (kzg/marshal.go:65-95): some verifying-key bytes, the marker, then
    unsafe.WriteSlice(pk.G1) = uint64 length + raw []G1Affine memory."""
    g = gm.G1Jac("bn254")
    o = oracle_mod.Oracle("bn254", "g1")
    n = 5000
    pts = g.generate_points(n, 31337, 5)
    vk = os.urandom(2 * 64 + 32)  # stands for Vk.WriteTo: the loader is told where the marker starts
    path = tmp_path / "srs.dump"
    with open(path, "wb") as f:
        f.write(vk)
        f.write(struct.pack("<Q", 0xDEADBEEF))
        f.write(struct.pack("<Q", n))
        f.write(pts.tobytes())
    sc = random_scalars(rng_for(9, n), g.curve, n)
    rb, err = g.register_bases_dump(path, offset=len(vk), check=2)
    assert err is None and rb.n == n
    jac, err = rb.MultiExp(sc)
    assert err is None and (g.jac_to_affine(jac) == o.msm_affine(pts, sc)).all()
    rb.release()
    # maxPkPoints (ReadDump's variadic limit)
    rb, err = g.register_bases_dump(path, offset=len(vk), max_points=1000)
    assert err is None and rb.n == 1000
    jac, err = rb.MultiExp(sc[:1000])
    assert (g.jac_to_affine(jac) == o.msm_affine(pts[:1000], sc[:1000])).all()
    assert rb.MultiExp(sc[:1001]) == (None, "len(points) != len(scalars)")
    rb.release()
    # without the marker, pointing at the length word
    rb, err = g.register_bases_dump(path, offset=len(vk) + 8, expect_marker=False)
    assert err is None and rb.n == n
    rb.release()
    # wrong position: marker mismatch, with the reference's text
    rb, err = g.register_bases_dump(path, offset=len(vk) + 1)
    assert rb is None and "marker mismatch" in err
    # a corrupted point is found by the optional validation, and ignored without it (ReadDump validates nothing)
    with open(path, "r+b") as f:
        f.seek(len(vk) + 16 + 64 * 123 + 40)
        f.write(b"\x01")
    rb, err = g.register_bases_dump(path, offset=len(vk), check=1)
    assert rb is None and "point 123" in err
    rb, err = g.register_bases_dump(path, offset=len(vk))
    assert err is None
    rb.release()
    # truncated file
    with open(path, "r+b") as f:
        f.truncate(len(vk) + 16 + 64 * 100)
    rb, err = g.register_bases_dump(path, offset=len(vk))
    assert rb is None and "short read" in err
