"""Host-side mirror of gnark-crypto's fr/fft package on top of the C ABI (include/gmsm.h, gmsm_fft_*).

Keeps the reference's names and meaning (ecc/bn254/fr/fft/domain.go, fft.go, bitreverse.go, options.go):

    d = NewDomain("bn254", m)                 # cardinality = next power of two >= m, Generator = fr.Generator(m)
    d.FFT(a, DIF)                             # natural order in, bit-reversed order out
    d.FFT(a, DIT, OnCoset())                  # bit-reversed in, natural out, evaluated on the coset u*<w>
    d.FFTInverse(a, DIF)
    BitReverse("bn254", a)

`a` is a numpy uint64 array (cardinality, fr_limbs) in the layout of []fr.Element (Montgomery limbs); transforms return
a new array (the C entry works in place on its copy).  The *_device variants take a raw device pointer and work in place.
"""
import ctypes

import numpy as np

from . import _lib
from .curves import CURVES

DIT, DIF = 0, 1  # fft.Decimation (fft.go:17-22)


class _Option:
    def __init__(self, coset=False):
        self.coset = coset


def OnCoset():
    """fft.OnCoset() (options.go): evaluate on / interpolate from the coset FrMultiplicativeGen * <Generator>."""
    return _Option(coset=True)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Domain:
    """fft.Domain (domain.go:24-60), resident on the device."""

    def __init__(self, curve, m):
        self.curve = CURVES[curve] if isinstance(curve, str) else curve
        self.gid = _lib.GROUP_IDS[(self.curve.name, "g1")]
        L = _lib.load()
        h = ctypes.c_uint64(0)
        rc = L.gmsm_fft_domain_new(self.gid, int(m), ctypes.byref(h))
        if rc:
            raise ValueError(_lib.last_error())  # NewDomain panics when the root of unity does not exist
        self.handle = h.value
        nl = self.curve.fr_limbs
        card = ctypes.c_uint64(0)
        vals = [np.zeros(nl, dtype=np.uint64) for _ in range(5)]
        L.gmsm_fft_domain_info(self.handle, ctypes.byref(card), *[_ptr(v) for v in vals])
        self.Cardinality = int(card.value)
        (self.Generator, self.GeneratorInv, self.CardinalityInv, self.FrMultiplicativeGen,
         self.FrMultiplicativeGenInv) = vals

    def _run(self, a, d_a, inverse, decimation, opts, stream=0):
        L = _lib.load()
        coset = any(o.coset for o in opts)
        if a is not None:
            a = np.array(a, dtype=np.uint64).reshape(-1, self.curve.fr_limbs)
            rc = L.gmsm_fft(self.handle, _ptr(a), None, a.shape[0], int(inverse), int(decimation), int(coset), None)
        else:
            rc = L.gmsm_fft(self.handle, None, d_a, self.Cardinality, int(inverse), int(decimation), int(coset), stream or None)
        if rc:
            raise ValueError(_lib.last_error())
        return a

    def FFT(self, a, decimation, *opts):
        return self._run(a, None, False, decimation, opts)

    def FFTInverse(self, a, decimation, *opts):
        return self._run(a, None, True, decimation, opts)

    def fft_device(self, d_a, decimation, *opts, inverse=False, stream=0):
        """In place on a device vector of Cardinality elements."""
        self._run(None, d_a, inverse, decimation, opts, stream)

    def release(self):
        if self.handle:
            _lib.load().gmsm_fft_domain_release(self.handle)
            self.handle = 0


def NewDomain(curve, m):
    return Domain(curve, m)


def BitReverse(curve, a=None, d_a=None, n=None, stream=0):
    """fft.BitReverse (bitreverse.go:20): len(a) must be a power of 2."""
    c = CURVES[curve] if isinstance(curve, str) else curve
    gid = _lib.GROUP_IDS[(c.name, "g1")]
    L = _lib.load()
    if a is not None:
        a = np.array(a, dtype=np.uint64).reshape(-1, c.fr_limbs)
        rc = L.gmsm_fft_bit_reverse(gid, _ptr(a), None, a.shape[0], None)
    else:
        rc = L.gmsm_fft_bit_reverse(gid, None, d_a, n, stream or None)
    if rc:
        raise ValueError(_lib.last_error())
    return a
