"""Pins the CPU oracle (oracle/msm_oracle.c) before anything is compared with it.  The Go reference cannot run here,
and its tests store no MSM outputs, so the pins are:
  1. field ops against Python big-int arithmetic (the reference's field tests do the same against math/big);
  2. the group law and a 2-term MSM against the RFC 9380 known-answer points stored in the reference's own
     hash_vectors_test.go (tests/golden/hash_vectors.json; generator script next to it);
  3. the reference's algebraic MSM identities (multiexp_test.go:54-60, 95-126, 128-182, 186-216) for every window size;
  4. an independent pure-Python affine MSM (oracle/pyref.py) on small random inputs;
  5. structural: the signed digits reconstruct the scalar (multiexp.go:749-800)."""
import json
import os

import numpy as np
import pytest

from conftest import ALL_GROUPS, int_to_limbs, random_field_limbs, random_scalars, rng_for, scalars_from_ints

HERE = os.path.dirname(os.path.abspath(__file__))


def _val(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bw6_761"])
@pytest.mark.parametrize("fld", ["fp", "fr"])
def test_field_ops_vs_bigint(oracle_mod, pyref_mod, curve, fld):
    c = pyref_mod.curves.CURVES[curve]
    q, n = (c.p, c.fp_limbs) if fld == "fp" else (c.r, c.fr_limbs)
    R = 1 << (64 * n)
    Rinv = pow(R, -1, q)
    F = oracle_mod.Field(f"{c.name}_{fld}", n)
    rng = rng_for(11, n, fld == "fp")
    edge = [0, 1, 2, q - 1, q - 2, R % q, R * R % q, (1 << 64) - 1, 1 << 64, q >> 1]
    vals = [np.array(int_to_limbs(v, n), dtype=np.uint64) for v in edge] + list(random_field_limbs(rng, q, n, 60))
    for a in vals:
        va = _val(a)
        assert _val(F.neg(a)) == (-va) % q
        assert _val(F.dbl(a)) == 2 * va % q
        assert _val(F.sqr(a)) == va * va * Rinv % q
        assert _val(F.from_mont(a)) == va * Rinv % q
        assert _val(F.to_mont(a)) == va * R % q
        if va:
            assert _val(F.mul(F.inv(a), a)) == R % q  # a^-1 * a = 1 (Montgomery one)
        for b in vals[::3]:
            vb = _val(b)
            assert _val(F.mul(a, b)) == va * vb * Rinv % q
            assert _val(F.add(a, b)) == (va + vb) % q
            assert _val(F.sub(a, b)) == (va - vb) % q
    assert _val(F.inv(vals[0])) == 0


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_e2_ops_vs_bigint(oracle_mod, pyref_mod, curve):
    c = pyref_mod.curves.CURVES[curve]
    n = c.fp_limbs
    F = oracle_mod.Field(f"{c.name}_e2", 2 * n)
    rng = rng_for(12, n)
    raw = random_field_limbs(rng, c.p, n, 40).reshape(20, 2 * n)
    raw[0] = 0
    raw[1, n:] = 0
    Rinv = pow(c.fp_R, -1, c.p)
    to_py = lambda l: pyref_mod.Fp2(_val(l[:n]) * Rinv, _val(l[n:]) * Rinv, c.p)
    eq = lambda l, z: (_val(l[:n]) * Rinv % c.p, _val(l[n:]) * Rinv % c.p) == (z.a0, z.a1)
    for a in raw:
        A = to_py(a)
        assert eq(F.sqr(a), A * A) and eq(F.neg(a), -A) and eq(F.dbl(a), A + A)
        if not A.is_zero():
            assert eq(F.mul(F.inv(a), a), pyref_mod.Fp2(1, 0, c.p))
        for b in raw[::4]:
            B = to_py(b)
            assert eq(F.mul(a, b), A * B) and eq(F.add(a, b), A + B) and eq(F.sub(a, b), A - B)


def _golden():
    return json.load(open(os.path.join(HERE, "golden", "hash_vectors.json")))


def _golden_points(pyref_mod, curve, which):
    c = pyref_mod.curves.CURVES[curve]
    g = pyref_mod.Group(c, which)

    def pt(v):
        if g.ext == 1:
            return (int(v[0][0], 16), int(v[1][0], 16))
        return (pyref_mod.Fp2(int(v[0][0], 16), int(v[0][1], 16), c.p), pyref_mod.Fp2(int(v[1][0], 16), int(v[1][1], 16), c.p))
    return g, [(pt(k["P"]), pt(k["Q0"]), pt(k["Q1"])) for k in _golden()[curve][which]]


H_EFF = {  # RFC 9380 effective cofactors: P = [h_eff](Q0 + Q1); BN254 G1 has cofactor 1
    ("bn254", "g1"): 1,
    ("bls12_381", "g1"): 0xD201000000010001,
    ("bls12_381", "g2"): 0xBC69F08F2EE75B3584C6A0EA91B352888E2A8E9145AD7689986FF031508FFE1329C2F178731DB956D82BF015D1212B02EC0EC69D7477C1AE954CBC06689F6A359894C0ADEBBF6B4E8020005AAA95551,
}


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2")])
def test_group_law_on_rfc9380_known_answers(oracle_mod, pyref_mod, curve, which):
    """Golden vectors of the reference (hash_vectors_test.go): Q0, Q1, P on the curve with P = [h_eff](Q0 + Q1)."""
    o = oracle_mod.Oracle(curve, which)
    g, cases = _golden_points(pyref_mod, curve, which)
    assert len(cases) == 5
    for P, Q0, Q1 in cases:
        assert g.on_curve(P) and g.on_curve(Q0) and g.on_curve(Q1)
        q0 = np.array(g.point_to_limbs(Q0), dtype=np.uint64)
        q1 = np.array(g.point_to_limbs(Q1), dtype=np.uint64)
        acc = o.xyzz_add_mixed(o.xyzz_add_mixed(o.xyzz_infinity(), q0), q1)
        s_aff = o.jac_to_affine(o.xyzz_to_jac(acc))
        assert g.point_from_limbs(s_aff) == g.add(Q0, Q1)
        # full XYZZ add and the doubling path: (Q0 + Q1) + (Q0 + Q1) == 2(Q0 + Q1)
        twice = o.jac_to_affine(o.xyzz_to_jac(o.xyzz_add(acc, acc)))
        assert (twice == o.jac_to_affine(o.xyzz_to_jac(o.xyzz_double(acc)))).all()
        assert g.point_from_limbs(twice) == g.add(g.add(Q0, Q1), g.add(Q0, Q1))
        h = H_EFF.get((curve, which))
        if h is None:
            continue  # BN254 G2: cofactor clearing is an endomorphism formula, no single published multiplier
        got = o.jac_to_affine(o.scalar_mul(s_aff, h))
        assert g.point_from_limbs(got) == P
        if h < g.c.r:  # the known answer as a 2-term MSM: h*Q0 + h*Q1 == P, for several window sizes
            sc = scalars_from_ints(g.c, [h, h])
            pts = np.stack([q0, q1])
            for c in (2, 5, 8, 13, 16):
                assert g.point_from_limbs(o.msm_affine(pts, sc, c=c)) == P
            assert g.point_from_limbs(o.msm_affine(pts, sc)) == P


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_sum_of_squares_identity_every_window(oracle_mod, pyref_mod, curve, which):
    """multiexp_test.go:95-126 + :54-60: MSM({i*G}, {i*mixer}) == mixer * n(n+1)(2n+1)/6 * G for all c, n = 73 and 30."""
    o = oracle_mod.Oracle(curve, which)
    g = pyref_mod.Group(o.curve, which)
    mixer = 0x2B1A09F8E7D6C5B4A39281706F5E4D3C2B1A0918273645566778899AABBCCDDE % o.curve.r
    for n, closed in ((73, 132349), (30, 9455)):
        pts = o.gen_points(n, 1, 1)
        assert g.point_from_limbs(pts[n - 1]) == g.mul(n, g.gen)
        sc = scalars_from_ints(o.curve, [(i + 1) * mixer for i in range(n)])
        expected = g.mul(closed * mixer % o.curve.r, g.gen)
        cs = range(2, 17) if n == 73 else (4, 9, 16)
        for c in cs:
            assert g.point_from_limbs(o.msm_affine(pts, sc, c=c)) == expected, (curve, which, c)
        for nb_tasks, ncpu in ((0, 8), (128, 8), (51, 8), (0, 192)):  # multiexp_test.go:63-86 split consistency
            err, jac = o.multiexp(pts, sc, nb_tasks=nb_tasks, num_cpu=ncpu, nthreads=2)
            assert err == 0 and g.jac_from_limbs(jac) == expected


@pytest.mark.parametrize("curve,which", ALL_GROUPS)
def test_infinity_and_zero_cases(oracle_mod, curve, which):
    o = oracle_mod.Oracle(curve, which)
    n = 40
    rng = rng_for(13, o.coord_limbs)
    pts = o.gen_points(n, 5, 3)
    sc = random_scalars(rng, o.curve, n)
    z = 2 * o.coord_limbs
    for c in (3, 8, 16):
        assert (o.msm_c(np.zeros_like(pts), sc, c)[z:] == 0).all()      # all-infinity points -> Z == 0 (:128-162)
        assert (o.msm_c(pts, np.zeros_like(sc), c)[z:] == 0).all()      # all-zero scalars  -> Z == 0 (:164-182)
    assert (o.msm_affine(np.zeros_like(pts), sc) == 0).all()
    err, jac = o.multiexp(pts[:0], sc[:0])
    assert err == 0 and (jac[z:] == 0).all()                            # n = 0 -> infinity
    assert o.multiexp(pts[:5], sc[:4])[0] == 1                          # len mismatch (multiexp.go:61-64)
    assert o.multiexp(pts, sc, nb_tasks=1025)[0] == 2                   # NbTasks > 1024 (multiexp.go:66-71)


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bw6_761", "g1")])
def test_against_independent_python_msm(oracle_mod, pyref_mod, curve, which):
    o = oracle_mod.Oracle(curve, which)
    g = pyref_mod.Group(o.curve, which)
    n = 24
    rng = rng_for(14, o.coord_limbs)
    pts = o.gen_points(n, int(rng.integers(1, 2**60)), int(rng.integers(1, 2**60)))
    sc = random_scalars(rng, o.curve, n)
    pts[3] = 0
    pts[9] = pts[8]; sc[9] = sc[8]          # duplicated pair -> doubling inside a bucket (multiexp_test.go:241-245)
    pts[11] = pts[10]
    sc[11] = np.array(pyref_mod.fr_to_mont(o.curve, o.curve.r - pyref_mod.fr_from_mont(o.curve, sc[10])), dtype=np.uint64)  # P*s + P*(-s)
    py_pts = [g.point_from_limbs(p) for p in pts]
    py_sc = [pyref_mod.fr_from_mont(o.curve, s) for s in sc]
    expected = g.msm(py_pts, py_sc)
    for c in (2, 4, 7, 11, 16):
        assert g.point_from_limbs(o.msm_affine(pts, sc, c=c)) == expected
    assert g.point_from_limbs(o.msm_affine(pts, sc)) == expected


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bw6_761"])
def test_signed_digits_reconstruct_scalar(oracle_mod, pyref_mod, curve):
    o = oracle_mod.Oracle(curve, "g1")
    c_ = o.curve
    rng = rng_for(15, c_.fr_limbs)
    sc = random_scalars(rng, c_, 200)
    sc[:6] = scalars_from_ints(c_, [0, 1, c_.r - 1, (1 << 16) - 1, 1 << 15, (1 << 64) - 1])
    vals = [pyref_mod.fr_from_mont(c_, s) for s in sc]
    for c in (2, 3, 5, 8, 10, 13, 16):
        d = o.partition_scalars(sc, c).astype(np.int64)
        nwin = d.shape[0]
        assert nwin == (c_.fr_bits + c - 1) // c
        signed = np.where(d & 1, -((d >> 1) + 1), d >> 1)       # decode: even e -> +e/2, odd e -> -(e>>1)-1
        for i, v in enumerate(vals):
            assert sum(int(signed[j, i]) << (c * j) for j in range(nwin)) == v
        assert (signed[:-1].max() <= (1 << (c - 1)) - 1) and (signed[:-1].min() >= -(1 << (c - 1)))
        assert (signed[-1] >= 0).all()                            # top window never borrows (:788-800)


def test_no_carry_product_equals_generic_cios(oracle_mod):
    """FPF(mul) (the reference's no-carry Mul, element_purego.go:46-213) against _mulGeneric (element.go:470-591) on
    random and edge operands, every prime field in scope."""
    import importlib
    curves = importlib.import_module("gnark-crypto_amd.curves")
    for c in curves.CURVES.values():
        for fname, mod, limbs in ((f"{c.name}_fp", c.p, c.fp_limbs), (f"{c.name}_fr", c.r, c.fr_limbs)):
            F = oracle_mod.Field(fname, limbs)
            rng = rng_for(41, limbs, mod & 0xFFFF)
            vals = random_field_limbs(rng, mod, limbs, 60)
            edge = [0, 1, 2, mod - 1, mod - 2, (1 << (64 * limbs)) % mod, (mod + 1) // 2]
            edge = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs)] for v in edge], dtype=np.uint64)
            ops = np.concatenate([vals, edge])
            for i in range(len(ops)):
                for j in (i, (i * 7 + 3) % len(ops), len(ops) - 1 - i):
                    assert (F.mul(ops[i], ops[j]) == F.mul_generic(ops[i], ops[j])).all()


@pytest.mark.parametrize("curve,which", [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bw6_761", "g1")])
def test_batch_affine_buckets_equal_jacobian_buckets(oracle_mod, curve, which):
    """multiexp_test.go:221-272: the batch-affine windows (c >= 10 with enough buckets hit) and the extended-Jacobian
    windows must produce the same point; duplicated pairs and s/-s pairs exercise the P+P / P-P conflict handling."""
    o = oracle_mod.Oracle(curve, which)
    n = 3000
    pts = o.gen_points(n, 11, 13, nthreads=4)
    sc = random_scalars(rng_for(55, n), o.curve, n)
    pts[10:60] = pts[1000:1050]   # the same points twice ...
    sc[10:60] = sc[1000:1050]     # ... with the same scalars: P + P inside one bucket
    pts[70] = 0
    try:
        for c in (10, 11, 13, 16):
            if curve == "bw6_761" and c not in (10, 16):
                continue
            oracle_mod.set_batch_affine(True)
            a = o.msm_affine(pts, sc, c=c, nthreads=4)
            oracle_mod.set_batch_affine(False)
            b = o.msm_affine(pts, sc, c=c, nthreads=4)
            assert (a == b).all(), c
        oracle_mod.set_batch_affine(True)
        a = o.msm_affine(pts, sc, nthreads=4)
        oracle_mod.set_batch_affine(False)
        assert (a == o.msm_affine(pts, sc, nthreads=4)).all()
    finally:
        oracle_mod.set_batch_affine(True)
