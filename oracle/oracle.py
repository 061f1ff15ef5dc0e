"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/liboracle.so (the C restatement of
gnark-crypto's MultiExp path).  May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product path (gnark-crypto_amd/) never imports it.

Arrays are numpy uint64 in the Go memory layout: points (n, 2*coord_limbs) = X then Y (each A0 then
A1 over Fp2), scalars (n, fr_limbs) Montgomery, Jacobian (3*coord_limbs,), XYZZ (4*coord_limbs,).
"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
curves = importlib.import_module("gnark-crypto_amd.curves")

_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("msm_oracle.c", "fp_tmpl.h", "e2_tmpl.h", "curve_tmpl.h", "fft_tmpl.h", "oracle_params.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_batch_affine(on):
    """True (default): windows use batch-affine buckets where the reference would (multiexp.go:232-294)."""
    lib().oracle_set_batch_affine(1 if on else 0)


BASELINE_NOTE = ("C restatement of gnark-crypto's algorithm: no-carry Montgomery product (MULX via gcc, no hand-written "
                 "ADX assembly), batch-affine buckets with the reference's thresholds for c >= 10, ext-Jacobian otherwise")


class Field:
    """Prime field (e.g. 'bn254_fp') or quadratic extension ('bn254_e2') ops on Montgomery limb arrays."""

    def __init__(self, name, limbs):
        self.name, self.limbs, self.L = name, limbs, lib()

    def _bin(self, op, a, b):
        z = np.zeros(self.limbs, dtype=np.uint64)
        getattr(self.L, f"oracle_{self.name}_{op}")(_p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(z))
        return z

    def _un(self, op, a):
        z = np.zeros(self.limbs, dtype=np.uint64)
        getattr(self.L, f"oracle_{self.name}_{op}")(_p(np.ascontiguousarray(a, dtype=np.uint64)), _p(z))
        return z

    def mul(self, a, b): return self._bin("mul", a, b)
    def mul_ns(self, iters=2_000_000):
        """nanoseconds per Montgomery product of this (prime) field on the current host, one core"""
        f = getattr(self.L, f"oracle_{self.name}_mul_ns")
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_uint]
        return float(f(iters))

    def mul_generic(self, a, b): return self._bin("mul_generic", a, b)  # _mulGeneric (CIOS with the overflow word)
    def add(self, a, b): return self._bin("add", a, b)
    def sub(self, a, b): return self._bin("sub", a, b)
    def neg(self, a): return self._un("neg", a)
    def dbl(self, a): return self._un("dbl", a)
    def sqr(self, a): return self._un("sqr", a)
    def inv(self, a): return self._un("inv", a)
    def dot(self, a, b):
        """sum_i a[i]*b[i] over (n, limbs) Montgomery arrays -> Montgomery limbs of the sum of products."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.limbs)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, self.limbs)
        assert a.shape == b.shape
        z = np.zeros(self.limbs, dtype=np.uint64)
        f = getattr(self.L, f"oracle_{self.name}_dot")
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        f(_p(a), _p(b), a.shape[0], _p(z))
        return z

    def from_mont(self, a): return self._un("from_mont", a)
    def to_mont(self, a): return self._un("to_mont", a)


DIT, DIF = 0, 1  # fft.Decimation (fr/fft/fft.go:17-22)


class FFT:
    """fr/fft of one curve's scalar field: FFT('bn254'). Arrays are (n, fr_limbs) uint64 Montgomery limbs."""

    def __init__(self, curve):
        self.curve = curves.CURVES[curve] if isinstance(curve, str) else curve
        self.name = f"{self.curve.name}_fr"
        self.limbs = self.curve.fr_limbs
        self.L = lib()

    def generator(self, m):
        """fr.Generator(m) (fr/generator.go:18): Montgomery limbs, or None when the root does not exist."""
        z = np.zeros(self.limbs, dtype=np.uint64)
        f = getattr(self.L, f"oracle_{self.name}_fft_generator")
        f.argtypes = [ctypes.c_uint64, ctypes.c_void_p]
        f.restype = ctypes.c_int
        return z if f(m, _p(z)) == 0 else None

    def transform(self, a, inverse=False, decimation=DIF, coset=False):
        """(*Domain).FFT / FFTInverse on a copy of a (len = cardinality)."""
        a = np.array(a, dtype=np.uint64).reshape(-1, self.limbs)
        f = getattr(self.L, f"oracle_{self.name}_fft")
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        f.restype = ctypes.c_int
        if f(_p(a), a.shape[0], int(inverse), int(decimation), int(coset)) != 0:
            raise ValueError("len(a) must be a power of two within the field's 2-adicity")
        return a

    def bit_reverse(self, a):
        a = np.array(a, dtype=np.uint64).reshape(-1, self.limbs)
        f = getattr(self.L, f"oracle_{self.name}_fft_bit_reverse")
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        f(_p(a), a.shape[0])
        return a


class Oracle:
    """One (curve, group): e.g. Oracle('bn254', 'g1')."""

    def __init__(self, curve, group):
        self.curve = curves.CURVES[curve] if isinstance(curve, str) else curve
        self.group = group
        self.name = f"{self.curve.name}_{group}"
        self.L = lib()
        ext = 1 if group == "g1" else self.curve.g2_ext
        self.ext = ext
        self.coord_limbs = self.curve.fp_limbs * ext
        self.fr_limbs = self.curve.fr_limbs
        self.aff_limbs = 2 * self.coord_limbs
        self.jac_limbs = 3 * self.coord_limbs
        self.xyzz_limbs = 4 * self.coord_limbs
        f = self._fn("multiexp")
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self._fn("msm_c").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
        self._fn("partition_scalars").argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p]
        self._fn("process_chunk").argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        self._fn("gen_points").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self._fn("scalar_mul").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._fn("xyzz_add_mixed").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self._fn("best_c").argtypes = [ctypes.c_size_t]
        self._fn("best_c").restype = ctypes.c_uint
        self._fn("nb_chunks").argtypes = [ctypes.c_uint]
        self._fn("nb_chunks").restype = ctypes.c_uint

    def _fn(self, op):
        return getattr(self.L, f"oracle_{self.name}_{op}")

    @property
    def generator(self):
        c = self.curve
        R = c.fp_R
        n = c.fp_limbs
        def mont(v):
            v = v * R % c.p
            return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]
        if self.group == "g1":
            vals = mont(c.g1[0]) + mont(c.g1[1])
        elif self.ext == 1:
            vals = mont(c.g2[0]) + mont(c.g2[1])
        else:
            (x0, x1), (y0, y1) = c.g2
            vals = mont(x0) + mont(x1) + mont(y0) + mont(y1)
        return np.array(vals, dtype=np.uint64)

    # ---- MSM
    def multiexp(self, points, scalars, nb_tasks=0, num_cpu=8, nthreads=1):
        """(*Jac).MultiExp semantics. Returns (err, jac) with err 0 / 1 / 2."""
        points = np.ascontiguousarray(points, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        err = self._fn("multiexp")(_p(points), points.shape[0] if points.ndim == 2 else points.size // self.aff_limbs,
                                    _p(scalars), scalars.shape[0] if scalars.ndim == 2 else scalars.size // self.fr_limbs,
                                    nb_tasks, num_cpu, nthreads, _p(out))
        return err, out

    def msm_c(self, points, scalars, c, nthreads=1):
        points = np.ascontiguousarray(points, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        n = points.size // self.aff_limbs
        assert scalars.size // self.fr_limbs == n
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        self._fn("msm_c")(_p(points), _p(scalars), n, c, nthreads, _p(out))
        return out

    def msm_affine(self, points, scalars, c=None, nthreads=1):
        """Canonical comparison value: affine (X,Y) Montgomery limbs of the MSM."""
        if c is None:
            err, jac = self.multiexp(points, scalars, nthreads=nthreads)
            assert err == 0
        else:
            jac = self.msm_c(points, scalars, c, nthreads)
        return self.jac_to_affine(jac)

    def jac_to_affine(self, jac):
        out = np.zeros(self.aff_limbs, dtype=np.uint64)
        self._fn("jac_to_affine")(_p(np.ascontiguousarray(jac, dtype=np.uint64)), _p(out))
        return out

    def nb_chunks(self, c):
        return int(self._fn("nb_chunks")(c))

    def best_c(self, n):
        return int(self._fn("best_c")(n))

    def partition_scalars(self, scalars, c):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        n = scalars.size // self.fr_limbs
        digits = np.zeros((self.nb_chunks(c), n), dtype=np.uint16)
        self._fn("partition_scalars")(_p(scalars), n, c, _p(digits))
        return digits

    def process_chunk(self, c, points, digits):
        points = np.ascontiguousarray(points, dtype=np.uint64)
        digits = np.ascontiguousarray(digits, dtype=np.uint16)
        out = np.zeros(self.xyzz_limbs, dtype=np.uint64)
        self._fn("process_chunk")(c, _p(points), _p(digits), digits.size, _p(out))
        return out

    # ---- group law (unit tests)
    def xyzz_infinity(self):
        out = np.zeros(self.xyzz_limbs, dtype=np.uint64)
        self._fn("xyzz_set_infinity")(_p(out))
        return out

    def xyzz_add_mixed(self, p, a, negate=False):
        p = np.array(p, dtype=np.uint64)
        self._fn("xyzz_add_mixed")(_p(p), _p(np.ascontiguousarray(a, dtype=np.uint64)), int(negate))
        return p

    def xyzz_add(self, p, q):
        p = np.array(p, dtype=np.uint64)
        self._fn("xyzz_add")(_p(p), _p(np.ascontiguousarray(q, dtype=np.uint64)))
        return p

    def xyzz_double(self, q):
        p = np.zeros(self.xyzz_limbs, dtype=np.uint64)
        self._fn("xyzz_double")(_p(p), _p(np.ascontiguousarray(q, dtype=np.uint64)))
        return p

    def xyzz_to_jac(self, p):
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        self._fn("xyzz_to_jac")(_p(np.ascontiguousarray(p, dtype=np.uint64)), _p(out))
        return out

    def jac_add(self, p, q):
        p = np.array(p, dtype=np.uint64)
        self._fn("jac_add_assign")(_p(p), _p(np.ascontiguousarray(q, dtype=np.uint64)))
        return p

    def scalar_mul(self, a, k):
        """[k]a with k a python int (plain, not Montgomery); returns Jacobian limbs."""
        nl = max(1, (k.bit_length() + 63) // 64)
        kl = np.array([(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        self._fn("scalar_mul")(_p(np.ascontiguousarray(a, dtype=np.uint64)), _p(kl), nl, _p(out))
        return out

    def fixed_base_msm_affine(self, a_mont, b_mont):
        """Affine limbs of [sum_i a_i b_i] G for Montgomery scalar arrays a, b: the closed form of
        MultiExp({[a_i]G}, {b_i}) (same shape as the reference's sum i^2 identity, multiexp_test.go:54-60)."""
        fr = Field(f"{self.curve.name}_fr", self.fr_limbs)
        k_limbs = fr.from_mont(fr.dot(a_mont, b_mont))
        k = sum(int(v) << (64 * i) for i, v in enumerate(k_limbs))
        return self.jac_to_affine(self.scalar_mul(self.generator, k))

    def gen_points(self, n, k0, k1, nthreads=1, base=None):
        """points[i] = [k0 + i*k1] * base  (base defaults to the group generator), affine Montgomery limbs."""
        base = self.generator if base is None else np.ascontiguousarray(base, dtype=np.uint64)
        nl = self.fr_limbs
        k0 %= self.curve.r
        k1 %= self.curve.r
        a0 = np.array([(k0 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)
        a1 = np.array([(k1 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)
        out = np.zeros((n, self.aff_limbs), dtype=np.uint64)
        if n:
            self._fn("gen_points")(_p(base), _p(a0), _p(a1), nl, n, nthreads, _p(out))
        return out
