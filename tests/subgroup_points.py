"""Points ON a curve but (almost surely) outside its r-torsion, and [r]P with the group order itself as the scalar:
helpers shared by the CPU model test (test_subgroup_model.py) and the GPU ingest tests."""


def sqrt_fp(a, p):
    r = pow(a, (p + 1) // 4, p)  # every base field in scope is 3 mod 4
    return r if r * r % p == a % p else None


def fp2_pow(pyref, a, e):
    r = pyref.Fp2(1, 0, a.p)
    while e:
        if e & 1:
            r = r * a
        a = a * a
        e >>= 1
    return r


def sqrt_fp2(pyref, a):
    """Square root in Fp[u]/(u^2+1), p = 3 mod 4 (complex method)."""
    p = a.p
    if a.is_zero():
        return a
    a1 = fp2_pow(pyref, a, (p - 3) // 4)
    alpha = a1 * a1 * a
    a0 = pyref.Fp2(alpha.a0, -alpha.a1, p) * alpha  # alpha^p * alpha
    if a0 == pyref.Fp2(-1, 0, p):
        return None
    x0 = a1 * a
    if alpha == pyref.Fp2(-1, 0, p):
        return pyref.Fp2(0, 1, p) * x0
    b = fp2_pow(pyref, pyref.Fp2(1, 0, p) + alpha, (p - 1) // 2)
    return b * x0


def curve_b(pyref, pg):
    if pg.ext == 2:
        return pyref._G2_B_FP2[pg.c.name](pg.p)
    return pg.c.b if pg.which == "g1" else pyref._G2_B_FP[pg.c.name]


def curve_points(pyref, pg, count, start=1):
    """`count` points of E(F) found by try-and-increment on x = start, start + 1, ... (x = t + u over Fp2)."""
    out = []
    t = start
    while len(out) < count:
        if pg.ext == 1:
            x = t
            y = sqrt_fp((x * x * x + curve_b(pyref, pg)) % pg.p, pg.p)
        else:
            x = pyref.Fp2(t, 1, pg.p)
            y = sqrt_fp2(pyref, x * x * x + curve_b(pyref, pg))
            if y is not None and not (y * y == x * x * x + curve_b(pyref, pg)):
                y = None
        t += 1
        if y is None:
            continue
        assert pg.on_curve((x, y))
        out.append((x, y))
    return out


def times_r(pg, P):
    """[r]P with the group order itself as the scalar (pyref.Group.mul reduces its scalar mod r)."""
    R = None
    for bit in bin(pg.c.r)[2:]:
        R = pg.add(R, R)
        if bit == "1":
            R = pg.add(R, P)
    return R


def curve_point_outside_subgroup(pyref, pg):
    """A point ON the curve whose order does not divide r (exists whenever the cofactor is not 1)."""
    for P in curve_points(pyref, pg, 200):
        if times_r(pg, P) is not None:
            return P
    raise AssertionError("no point outside the subgroup found")
