"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Independent pure-Python big-int model of the curves in scope: affine arithmetic over Fp / Fp2 with Python
ints, double-and-add scalar multiplication and a naive MSM.  Shares no code with the C oracle
(oracle/msm_oracle.c) or the HIP kernels; it exists to guard against shared mistakes on small inputs
 and to convert between integers and the reference's Montgomery limb layout
(ecc/bn254/fp/element.go:24-36: little-endian uint64 limbs of x*R mod q).
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
curves = importlib.import_module("gnark-crypto_amd.curves")


# ---------------------------------------------------------------- field element <-> limbs
def to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def fp_to_mont(c, v):
    return to_limbs(v % c.p * c.fp_R % c.p, c.fp_limbs)


def fp_from_mont(c, limbs):
    return from_limbs(limbs) * pow(c.fp_R, -1, c.p) % c.p


def fr_to_mont(c, v):
    return to_limbs(v % c.r * c.fr_R % c.r, c.fr_limbs)


def fr_from_mont(c, limbs):
    return from_limbs(limbs) * pow(c.fr_R, -1, c.r) % c.r


# ---------------------------------------------------------------- Fp / Fp2 as python values
class Fp2:
    """a0 + a1*u, u^2 = -1 (ecc/bn254/internal/fptower/e2.go:14-16)."""
    __slots__ = ("a0", "a1", "p")

    def __init__(self, a0, a1, p):
        self.a0, self.a1, self.p = a0 % p, a1 % p, p

    def __add__(self, o): return Fp2(self.a0 + o.a0, self.a1 + o.a1, self.p)
    def __sub__(self, o): return Fp2(self.a0 - o.a0, self.a1 - o.a1, self.p)
    def __neg__(self): return Fp2(-self.a0, -self.a1, self.p)
    def __mul__(self, o):
        if isinstance(o, int):
            return Fp2(self.a0 * o, self.a1 * o, self.p)
        return Fp2(self.a0 * o.a0 - self.a1 * o.a1, self.a0 * o.a1 + self.a1 * o.a0, self.p)
    def __eq__(self, o): return self.a0 == o.a0 and self.a1 == o.a1
    def inv(self):
        n = pow(self.a0 * self.a0 + self.a1 * self.a1, -1, self.p)
        return Fp2(self.a0 * n, -self.a1 * n, self.p)
    def is_zero(self): return self.a0 == 0 and self.a1 == 0
    def __repr__(self): return f"Fp2({self.a0:#x},{self.a1:#x})"


class Group:
    """Affine short-Weierstrass group y^2 = x^3 + b (a = 0) over Fp (ext=1) or Fp2 (ext=2). None = infinity."""

    def __init__(self, curve, which):
        self.c = curve
        self.which = which
        self.ext = 1 if which == "g1" else curve.g2_ext
        self.p = curve.p
        if which == "g1":
            self.gen = curve.g1
        elif self.ext == 1:
            self.gen = curve.g2
        else:
            (x0, x1), (y0, y1) = curve.g2
            self.gen = (Fp2(x0, x1, self.p), Fp2(y0, y1, self.p))

    @property
    def coord_limbs(self):
        return self.c.fp_limbs * self.ext

    # field helpers
    def _inv(self, v):
        return pow(v, -1, self.p) if self.ext == 1 else v.inv()

    def _red(self, v):
        return v % self.p if self.ext == 1 else v

    def neg(self, P):
        if P is None:
            return None
        x, y = P
        return (x, self._red(-y) if self.ext == 1 else -y)

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if self._red(y1 + y2) == 0 if self.ext == 1 else (y1 + y2).is_zero():
                return None
            lam = (x1 * x1 * 3) * self._inv(y1 * 2) if self.ext == 2 else 3 * x1 * x1 * self._inv(2 * y1 % self.p)
        else:
            lam = (y2 - y1) * self._inv(self._red(x2 - x1) if self.ext == 1 else (x2 - x1))
        x3 = self._red(lam * lam - x1 - x2)
        y3 = self._red(lam * (x1 - x3) - y1)
        return (x3, y3)

    def mul(self, k, P):
        k %= self.c.r if k >= 0 else self.c.r
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.add(R, R)
            if bit == "1":
                R = self.add(R, P)
        return R

    def msm(self, points, scalars):
        acc = None
        for P, s in zip(points, scalars):
            if P is None or s % self.c.r == 0:
                continue
            acc = self.add(acc, self.mul(s % self.c.r, P))
        return acc

    # limb (de)serialisation in the Go memory layout: X then Y, each A0 then A1 for Fp2; (0,0) = infinity
    def point_to_limbs(self, P):
        n = self.c.fp_limbs
        if P is None:
            return [0] * (2 * n * self.ext)
        out = []
        for coord in P:
            if self.ext == 1:
                out += fp_to_mont(self.c, coord)
            else:
                out += fp_to_mont(self.c, coord.a0) + fp_to_mont(self.c, coord.a1)
        return out

    def point_from_limbs(self, limbs):
        n = self.c.fp_limbs
        limbs = [int(v) for v in limbs]
        if all(v == 0 for v in limbs):
            return None
        vals = [fp_from_mont(self.c, limbs[i * n:(i + 1) * n]) for i in range(2 * self.ext)]
        if self.ext == 1:
            return (vals[0], vals[1])
        return (Fp2(vals[0], vals[1], self.p), Fp2(vals[2], vals[3], self.p))

    def jac_from_limbs(self, limbs):
        """Jacobian (X,Y,Z) limbs -> affine python point."""
        n = self.c.fp_limbs
        limbs = [int(v) for v in limbs]
        vals = [fp_from_mont(self.c, limbs[i * n:(i + 1) * n]) for i in range(3 * self.ext)]
        if self.ext == 1:
            X, Y, Z = vals
            if Z == 0:
                return None
            zi = pow(Z, -1, self.p)
            return (X * zi * zi % self.p, Y * zi * zi * zi % self.p)
        X, Y, Z = (Fp2(vals[2 * i], vals[2 * i + 1], self.p) for i in range(3))
        if Z.is_zero():
            return None
        zi = Z.inv()
        return (X * zi * zi, Y * zi * zi * zi)

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        if self.ext == 1:
            b = self.c.b if self.which == "g1" else _G2_B_FP[self.c.name]
            return (y * y - x * x * x - b) % self.p == 0
        b = _G2_B_FP2[self.c.name](self.p)
        return (y * y - x * x * x - b).is_zero()


# twist coefficients (only used to sanity-check generated test inputs):
#   BN254 D-twist b' = 3/(9+u) (bn254.go:104-108); BLS12-381 M-twist b' = 4(1+u) (bls12-381.go:99-103);
#   BW6-761 M-twist b' = 4 over Fp (bw6-761.go:93-95)
_G2_B_FP = {"bw6_761": 4}
_G2_B_FP2 = {
    "bn254": lambda p: Fp2(9, 1, p).inv() * 3,
    "bls12_381": lambda p: Fp2(4, 4, p),
}


# ---- IsInSubGroup as the reference computes it: one endomorphism identity per group (affine big-int model) ----
def _mul_plain(pg, k, P):
    """[k]P by double-and-add WITHOUT reducing k mod r (P need not be in the r-torsion)."""
    R = None
    for bit in bin(k)[2:] if k else "":
        R = pg.add(R, R)
        if bit == "1":
            R = pg.add(R, P)
    return R


def endo_phi(pg, P):
    """phi(x, y) = (w x, y), w = thirdRootOneG1 on G1 and its square on G2 (bls12-381/g1.go:530-534, bw6-761/g2.go:540-544)."""
    if P is None:
        return None
    w = pg.c.third_root_one_g1 if pg.which == "g1" else pg.c.third_root_one_g1 ** 2 % pg.p
    return (P[0] * w % pg.p, P[1]) if pg.ext == 1 else (P[0] * Fp2(w, 0, pg.p), P[1])


def endo_psi(pg, P):
    """psi(x, y) = (conj(x) u, conj(y) v): untwist-Frobenius-twist on G2 over Fp2 (bn254/g2.go:534-540)."""
    if P is None:
        return None
    u, v = Fp2(*pg.c.endo_u, pg.p), Fp2(*pg.c.endo_v, pg.p)
    conj = lambda a: Fp2(a.a0, -a.a1, pg.p)
    return (conj(P[0]) * u, conj(P[1]) * v)


def is_in_subgroup_endo(pg, P):
    """The predicate of (*G1Jac).IsInSubGroup / (*G2Jac).IsInSubGroup for a point ON the curve, identity for identity:
    bn254/g1.go:475-482 (prime order), bn254/g2.go:483-497, bls12-381/g1.go:481-492, bls12-381/g2.go:484-491,
    bw6-761/g1.go:482-496, bw6-761/g2.go:488-502."""
    x = pg.c.x_gen
    name, which = pg.c.name, pg.which
    if P is None:
        return True
    if name == "bn254" and which == "g1":
        return True
    if name == "bls12_381" and which == "g1":
        res = _mul_plain(pg, x, _mul_plain(pg, x, endo_phi(pg, P)))
        return pg.add(res, P) is None
    if name == "bls12_381":
        return pg.add(_mul_plain(pg, x, P), endo_psi(pg, P)) is None
    if name == "bn254":
        a = _mul_plain(pg, x, P)
        b = endo_psi(pg, a)
        a = pg.add(a, P)
        res = endo_psi(pg, b)
        c = pg.add(pg.add(res, b), a)
        res = endo_psi(pg, res)
        res = pg.add(pg.add(res, res), pg.neg(c))
        return res is None
    # BW6-761, both groups
    phip = endo_phi(pg, P)
    res = pg.add(_mul_plain(pg, x, phip), pg.neg(phip))
    res = _mul_plain(pg, x, _mul_plain(pg, x, res))
    res = pg.add(res, phip)
    t = pg.add(pg.add(_mul_plain(pg, x, P), P), res)
    return t is None
