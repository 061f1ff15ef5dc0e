"""Host-side mirror of gnark-crypto's MultiExp API on top of the C ABI (include/gmsm.h).

The reference's host language is Go; no Go toolchain exists in this environment, so the mirror above the C ABI is
Python (ctypes), keeping the reference's names, argument meaning and error behaviour
(ecc/bn254/multiexp.go:20-71, ecc/ecc.go:107-110):

    cfg = MultiExpConfig(NbTasks=0)
    p, err = G1Affine("bn254").MultiExp(points, scalars, cfg)      # -> (affine limbs, None) or (None, error string)
    j, err = G1Jac("bn254").MultiExp(points, scalars, cfg)         # -> Jacobian limbs

`points` is a numpy uint64 array (n, 2*coord_limbs) in the memory layout of Go's []G1Affine / []G2Affine and
`scalars` is (n, fr_limbs) in the layout of []fr.Element (both Montgomery form), i.e. exactly what a cgo caller passes
with unsafe.Pointer(&points[0]).  The cgo stub a maintainer would add is in INTEGRATION.md.
"""
from dataclasses import dataclass

import numpy as np

from . import _lib
from .curves import CURVES

ERR_LEN = "len(points) != len(scalars)"                      # multiexp.go:63
ERR_NBTASKS = "invalid config: config.NbTasks > 1024"        # multiexp.go:70


@dataclass
class MultiExpConfig:
    """ecc.MultiExpConfig (ecc/ecc.go:107-110). NbTasks is validated like the reference and otherwise ignored: the
    GPU grid replaces the goroutine pool."""
    NbTasks: int = 0


def _ptr(a):
    return a.ctypes.data_as(_lib.ctypes.c_void_p)


SHARD_MODES = {"auto": 0, "points": 1, "windows": 2}


def _dev_array(devices):
    """list of device indices (one per logical rank) -> (int*, count) for the C ABI; None -> (NULL, 0)."""
    if not devices:
        return None, 0
    arr = (_lib.ctypes.c_int * len(devices))(*[int(d) for d in devices])
    return arr, len(devices)


def set_devices(devices=None):
    """gmsm_set_devices: the devices the drop-in entries spread a MultiExp over (None / [] = every visible device)."""
    dev, nd = _dev_array(devices)
    rc = _lib.load().gmsm_set_devices(dev, nd)
    if rc:
        raise RuntimeError("gmsm: " + _lib.last_error())


def get_devices():
    L = _lib.load()
    buf = (_lib.ctypes.c_int * 64)()
    n = L.gmsm_get_devices(buf, 64)
    return [buf[i] for i in range(min(n, 64))]


class _Group:
    group = None  # "g1" / "g2"

    def __init__(self, curve="bn254"):
        self.curve = CURVES[curve] if isinstance(curve, str) else curve
        self.gid = _lib.GROUP_IDS[(self.curve.name, self.group)]
        ext = 1 if self.group == "g1" else self.curve.g2_ext
        self.coord_limbs = self.curve.fp_limbs * ext
        self.fr_limbs = self.curve.fr_limbs
        self.aff_limbs = 2 * self.coord_limbs
        self.jac_limbs = 3 * self.coord_limbs
        self.xyzz_limbs = 4 * self.coord_limbs

    def _check(self, points, scalars):
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, self.fr_limbs)
        return points, scalars

    def _error(self, rc):
        if rc == _lib.GMSM_ERR_LEN:
            return ERR_LEN
        if rc == _lib.GMSM_ERR_CONFIG:
            return ERR_NBTASKS
        return "gmsm: " + _lib.last_error()

    def _multiexp_jac(self, points, scalars, config):
        L = _lib.load()
        points, scalars = self._check(points, scalars)
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp(self.gid, _ptr(points), points.shape[0], _ptr(scalars), scalars.shape[0],
                             int(config.NbTasks), _ptr(out))
        return (out, None) if rc == 0 else (None, self._error(rc))

    def MultiExpSharded(self, points, scalars, config=MultiExpConfig(), devices=None, mode="auto"):
        """gmsm_multiexp_sharded: the same MultiExp spread over several devices inside the library (one host thread per
        logical rank; `devices` = one entry per rank, None = the configured list). Returns (jacobian_limbs, None) or
        (None, error)."""
        L = _lib.load()
        points, scalars = self._check(points, scalars)
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        dev, nd = _dev_array(devices)
        rc = L.gmsm_multiexp_sharded(self.gid, _ptr(points), points.shape[0], _ptr(scalars), scalars.shape[0],
                                     int(config.NbTasks), dev, nd, SHARD_MODES[mode], _ptr(out))
        return (out, None) if rc == 0 else (None, self._error(rc))

    def register_bases_sharded(self, points, devices=None):
        """gmsm_bases_register_sharded: full copies of the bases on every distinct device of the list."""
        L = _lib.load()
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
        handle = _lib.ctypes.c_uint64(0)
        dev, nd = _dev_array(devices)
        rc = L.gmsm_bases_register_sharded(self.gid, _ptr(points), points.shape[0], dev, nd, _lib.ctypes.byref(handle))
        if rc:
            raise RuntimeError(self._error(rc))
        return ResidentBases(self, handle.value, points.shape[0])

    def _fold_jac(self, points, combination_coeff, config):
        """(*G1Jac).Fold (multiexp.go:331): sum_i points[i] * coeff^i; returns (jacobian_limbs, None) or (None, error)."""
        L = _lib.load()
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
        coeff = np.ascontiguousarray(combination_coeff, dtype=np.uint64).reshape(self.fr_limbs)
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_fold(self.gid, _ptr(points), points.shape[0], _ptr(coeff), int(config.NbTasks), _ptr(out))
        return (out, None) if rc == 0 else (None, self._error(rc))

    def Fold(self, points, combination_coeff, config=MultiExpConfig()):
        """Fold of the Jac types returns Jacobian limbs, of the Affine types affine limbs (multiexp.go:320-340)."""
        jac, err = self._fold_jac(points, combination_coeff, config)
        if err is not None:
            return None, err
        return (self.jac_to_affine(jac) if type(self).__name__.endswith("Affine") else jac), None

    def _multiexp_affine(self, points, scalars, config):
        L = _lib.load()
        points, scalars = self._check(points, scalars)
        out = np.zeros(self.aff_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_affine(self.gid, _ptr(points), points.shape[0], _ptr(scalars), scalars.shape[0],
                                    int(config.NbTasks), _ptr(out))
        return (out, None) if rc == 0 else (None, self._error(rc))

    # ---- utilities
    @property
    def generator(self):
        """Affine generator of the group in Montgomery limbs (bn254.go:110-123 etc.)."""
        c = self.curve
        mont = lambda v: [(v * c.fp_R % c.p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(c.fp_limbs)]
        if self.group == "g1":
            vals = mont(c.g1[0]) + mont(c.g1[1])
        elif c.g2_ext == 1:
            vals = mont(c.g2[0]) + mont(c.g2[1])
        else:
            (x0, x1), (y0, y1) = c.g2
            vals = mont(x0) + mont(x1) + mont(y0) + mont(y1)
        return np.array(vals, dtype=np.uint64)

    def generate_points(self, n, k0, k1, nthreads=None, base=None):
        """points[i] = [k0 + i*k1] * base (default: the generator): on-curve, distinct, r-torsion synthetic bases."""
        import os
        L = _lib.load()
        base = self.generator if base is None else np.ascontiguousarray(base, dtype=np.uint64)
        nl = self.fr_limbs
        limbs = lambda v: np.array([(v % self.curve.r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)
        a0, a1 = limbs(k0), limbs(k1)
        out = np.zeros((n, self.aff_limbs), dtype=np.uint64)
        rc = L.gmsm_generate_points(self.gid, _ptr(base), _ptr(a0), _ptr(a1), nl, n, nthreads or 2 * _lib.effective_cpus(), _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    # ---- device-resident path (pointers are raw device addresses, e.g. torch.Tensor.data_ptr())
    def multiexp_device(self, d_points, d_scalars, n, stream=0):
        L = _lib.load()
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_device(self.gid, d_points, d_scalars, n, stream or None, _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    # ---- resident bases (device-resident SRS): register once, then MultiExp over any prefix with scalars only
    def BatchScalarMultiplication(self, base, scalars):
        """BatchScalarMultiplicationG1/G2 (ecc/bn254/g1.go:1039): affine scalars[i] * base for every i."""
        L = _lib.load()
        base = np.ascontiguousarray(base, dtype=np.uint64).reshape(self.aff_limbs)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, self.fr_limbs)
        out = np.zeros((scalars.shape[0], self.aff_limbs), dtype=np.uint64)
        rc = L.gmsm_batch_scalar_mul(self.gid, _ptr(base), _ptr(scalars), scalars.shape[0], _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    def batch_scalar_mul_device(self, base, d_scalars, n, d_out, stream=0):
        """Same with scalars and results in device memory (d_out: n affine points, Go layout)."""
        L = _lib.load()
        base = np.ascontiguousarray(base, dtype=np.uint64).reshape(self.aff_limbs)
        rc = L.gmsm_batch_scalar_mul_device(self.gid, _ptr(base), d_scalars, n, stream or None, d_out)
        if rc:
            raise RuntimeError(self._error(rc))

    def BatchJacobianToAffine(self, jac_points):
        """BatchJacobianToAffineG1 (ecc/bn254/g1.go:989): n Jacobian points -> n affine points, Z = 0 -> (0, 0)."""
        L = _lib.load()
        jac_points = np.ascontiguousarray(jac_points, dtype=np.uint64).reshape(-1, self.jac_limbs)
        out = np.zeros((jac_points.shape[0], self.aff_limbs), dtype=np.uint64)
        rc = L.gmsm_batch_jac_to_affine(self.gid, _ptr(jac_points), jac_points.shape[0], _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    def register_bases(self, points=None, d_points=None, n=None):
        """Upload (host `points`) or adopt (`d_points` device pointer, n points) the bases; returns a ResidentBases."""
        L = _lib.load()
        handle = _lib.ctypes.c_uint64(0)
        if points is not None:
            points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
            rc = L.gmsm_bases_register(self.gid, _ptr(points), None, points.shape[0], _lib.ctypes.byref(handle))
            n = points.shape[0]
        else:
            rc = L.gmsm_bases_register(self.gid, None, d_points, n, _lib.ctypes.byref(handle))
        if rc:
            raise RuntimeError(self._error(rc))
        return ResidentBases(self, handle.value, n)

    # ---- point ingest: wire format / raw dumps -> validated limbs, on the device
    @property
    def raw_point_bytes(self):
        """SizeOfG1AffineUncompressed / SizeOfG2AffineUncompressed (marshal.go): equals the in-memory size."""
        return 8 * self.aff_limbs

    def DecodeRaw(self, buf, subgroup_check=True, on_curve_check=True):
        """n uncompressed points (the bytes of RawBytes(), ecc/bn254/marshal.go:826) -> ((n, aff_limbs) Montgomery limbs,
        None) or (None, error text); the Decoder's checks (marshal.go:250-275) run on the device.  The failing index is
        the number after 'point' in the error text."""
        L = _lib.load()
        raw = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if raw.size % self.raw_point_bytes:
            return None, "short buffer"  # io.ErrShortBuffer
        n = raw.size // self.raw_point_bytes
        out = np.zeros((n, self.aff_limbs), dtype=np.uint64)
        bad = _lib.ctypes.c_int64(-1)
        check = 2 if subgroup_check else (1 if on_curve_check else 0)
        rc = L.gmsm_points_from_raw(self.gid, _ptr(raw), n, check, _ptr(out), None, _lib.ctypes.byref(bad))
        return (out, None) if rc == 0 else (None, self._error(rc))

    @property
    def compressed_point_bytes(self):
        """SizeOfG1AffineCompressed / SizeOfG2AffineCompressed (marshal.go): one coordinate."""
        return 4 * self.aff_limbs

    def DecodeCompressed(self, buf, subgroup_check=True, d_out=None):
        """n compressed points (the bytes of Bytes(), the Encoder's default: ecc/bn254/marshal.go:801-823) -> ((n, aff_limbs)
        Montgomery limbs, None) or (None, error text): setBytes' compressed branch (marshal.go:907-948) with Y = sqrt(X^3 + b)
        computed on the device.  d_out: leave the points in HBM instead (returns (n, None))."""
        L = _lib.load()
        comp = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if comp.size % self.compressed_point_bytes:
            return None, "short buffer"  # io.ErrShortBuffer
        n = comp.size // self.compressed_point_bytes
        bad = _lib.ctypes.c_int64(-1)
        check = 2 if subgroup_check else 0
        if d_out is not None:
            rc = L.gmsm_points_from_compressed(self.gid, _ptr(comp), n, check, None, d_out, _lib.ctypes.byref(bad))
            return (n, None) if rc == 0 else (None, self._error(rc))
        out = np.zeros((n, self.aff_limbs), dtype=np.uint64)
        rc = L.gmsm_points_from_compressed(self.gid, _ptr(comp), n, check, _ptr(out), None, _lib.ctypes.byref(bad))
        return (out, None) if rc == 0 else (None, self._error(rc))

    def Compress(self, points=None, d_points=None, n=None):
        """Bytes() of every point of a vector (host limbs or device pointer): (n * compressed_point_bytes uint8 array, None)
        or (None, error text)."""
        L = _lib.load()
        if points is not None:
            points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
            n = points.shape[0]
        out = np.zeros(n * self.compressed_point_bytes, dtype=np.uint8)
        rc = L.gmsm_points_compress(self.gid, _ptr(points) if points is not None else None, d_points, n, _ptr(out))
        return (out, None) if rc == 0 else (None, self._error(rc))

    def register_bases_compressed(self, buf, subgroup_check=True):
        """Decode (compressed) + validate + register in one call: returns (ResidentBases, None) or (None, error text)."""
        L = _lib.load()
        comp = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if comp.size % self.compressed_point_bytes:
            return None, "short buffer"
        n = comp.size // self.compressed_point_bytes
        handle = _lib.ctypes.c_uint64(0)
        bad = _lib.ctypes.c_int64(-1)
        rc = L.gmsm_bases_register_compressed(self.gid, _ptr(comp), n, 2 if subgroup_check else 0, _lib.ctypes.byref(handle),
                                              _lib.ctypes.byref(bad))
        return (ResidentBases(self, handle.value, n), None) if rc == 0 else (None, self._error(rc))

    def DecodeSlice(self, buf, subgroup_check=True):
        """What Decoder.Decode(&[]G1Affine) reads (marshal.go:220-277): uint32 big-endian length, then the points - all compressed
        (the Encoder's default, marshal.go:445-585) or all raw (RawEncoding(), :418, :586-640); the first point's flag bits say
        which (isCompressed, marshal.go:380-383)."""
        b = bytes(buf)
        if len(b) < 4:
            return None, "short buffer"
        n = int.from_bytes(b[:4], "big")
        if n == 0:
            return np.zeros((0, self.aff_limbs), dtype=np.uint64), None
        if len(b) < 5:
            return None, "short buffer"
        flag_bits = 2 if self.curve.name == "bn254" else 3
        flag = b[4] >> (8 - flag_bits)
        uncompressed = flag == 0 or (flag_bits == 3 and flag == 0b010)
        size = self.raw_point_bytes if uncompressed else self.compressed_point_bytes
        if len(b) < 4 + n * size:
            return None, "short buffer"
        body = b[4:4 + n * size]
        return self.DecodeRaw(body, subgroup_check) if uncompressed else self.DecodeCompressed(body, subgroup_check)

    def ValidatePoints(self, points=None, d_points=None, n=None, subgroup_check=True, by_definition=False):
        """IsOnCurve / IsInSubGroup over a whole vector of limb-form points (host array or device pointer): returns
        (True, None) or (False, error text).  by_definition: decide membership as [r]P = infinity (check level 3)
        instead of through the reference's endomorphism identities (level 2) - the same predicate, the cross-check."""
        L = _lib.load()
        bad = _lib.ctypes.c_int64(-1)
        level = (3 if by_definition else 2) if subgroup_check else 1
        if points is not None:
            points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.aff_limbs)
            rc = L.gmsm_points_validate(self.gid, _ptr(points), None, points.shape[0], level, _lib.ctypes.byref(bad))
        else:
            rc = L.gmsm_points_validate(self.gid, None, d_points, n, level, _lib.ctypes.byref(bad))
        return (True, None) if rc == 0 else (False, self._error(rc))

    def register_bases_raw(self, buf, subgroup_check=True):
        """Decode + validate + register in one call: returns (ResidentBases, None) or (None, error text)."""
        L = _lib.load()
        raw = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if raw.size % self.raw_point_bytes:
            return None, "short buffer"
        n = raw.size // self.raw_point_bytes
        handle = _lib.ctypes.c_uint64(0)
        bad = _lib.ctypes.c_int64(-1)
        rc = L.gmsm_bases_register_raw(self.gid, _ptr(raw), n, 2 if subgroup_check else 0, _lib.ctypes.byref(handle),
                                       _lib.ctypes.byref(bad))
        return (ResidentBases(self, handle.value, n), None) if rc == 0 else (None, self._error(rc))

    def register_bases_dump(self, path, offset=0, expect_marker=True, max_points=0, check=0):
        """kzg.SRS.ReadDump's point part (ecc/bn254/kzg/marshal.go:98-113; utils/unsafe/dump_slice.go:35-76) straight
        into HBM: marker, uint64 length, raw []G1Affine memory.  Returns (ResidentBases, None) or (None, error text)."""
        L = _lib.load()
        handle = _lib.ctypes.c_uint64(0)
        n = _lib.ctypes.c_size_t(0)
        bad = _lib.ctypes.c_int64(-1)
        rc = L.gmsm_bases_register_dump(self.gid, str(path).encode(), offset, 1 if expect_marker else 0, max_points, check,
                                        _lib.ctypes.byref(handle), _lib.ctypes.byref(n), _lib.ctypes.byref(bad))
        return (ResidentBases(self, handle.value, n.value), None) if rc == 0 else (None, self._error(rc))

    def default_window_bits(self, n):
        return int(_lib.load().gmsm_default_window_bits(self.gid, n))

    def default_plan(self, n):
        """What a MultiExp over n bases taken anew runs as: dict(window_bits, windows, entries_per_point, fused)."""
        v = [_lib.ctypes.c_uint(0) for _ in range(4)]
        rc = _lib.load().gmsm_default_plan(self.gid, n, *[_lib.ctypes.byref(x) for x in v])
        if rc:
            raise RuntimeError(self._error(rc))
        return {"window_bits": v[0].value, "windows": v[1].value, "entries_per_point": v[2].value, "fused": bool(v[3].value)}

    def num_windows(self, c):
        return int(_lib.load().gmsm_num_windows(self.gid, c))

    def window_sums_device(self, d_points, d_scalars, n, c, win_first=0, win_stride=1, stream=0):
        """XYZZ totals of windows win_first, win_first+win_stride, ... (window sharding across GPUs)."""
        L = _lib.load()
        nwin = self.num_windows(c)
        nloc = max(0, (nwin - win_first + win_stride - 1) // win_stride) if win_first < nwin else 0
        out = np.zeros((nloc, self.xyzz_limbs), dtype=np.uint64)
        rc = L.gmsm_window_sums_device(self.gid, d_points, d_scalars, n, c, win_first, win_stride, stream or None, _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    def window_sums_enqueue(self, d_points, d_scalars, n, c, win_first, win_stride, stream, d_out, bases=None):
        """gmsm_window_sums_enqueue: totals go to the device buffer d_out in stream order; returns without waiting."""
        L = _lib.load()
        rc = L.gmsm_window_sums_enqueue(self.gid, d_points, bases.handle if bases is not None else 0, d_scalars, n, c,
                                        win_first, win_stride, stream or None, d_out)
        if rc:
            raise RuntimeError(self._error(rc))

    def fold_window_sets(self, xyzz_sets, c):
        """gmsm_fold_window_sets: (nsets, nwin, xyzz_limbs) totals of point slices -> Jacobian result."""
        L = _lib.load()
        xyzz_sets = np.ascontiguousarray(xyzz_sets, dtype=np.uint64)
        nwin = self.num_windows(c)
        nsets = xyzz_sets.size // (nwin * self.xyzz_limbs)
        assert nsets >= 1 and xyzz_sets.size == nsets * nwin * self.xyzz_limbs
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_fold_window_sets(self.gid, c, _ptr(xyzz_sets), nsets, _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    def fold_windows(self, xyzz_windows, c):
        L = _lib.load()
        xyzz_windows = np.ascontiguousarray(xyzz_windows, dtype=np.uint64)
        assert xyzz_windows.size == self.num_windows(c) * self.xyzz_limbs
        out = np.zeros(self.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_fold_windows(self.gid, c, _ptr(xyzz_windows), _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out

    def jac_to_affine(self, jac):
        L = _lib.load()
        jac = np.ascontiguousarray(jac, dtype=np.uint64)
        out = np.zeros(self.aff_limbs, dtype=np.uint64)
        rc = L.gmsm_jac_to_affine(self.gid, _ptr(jac), _ptr(out))
        if rc:
            raise RuntimeError(self._error(rc))
        return out


class ResidentBases:
    """Bases kept on the device in the engine's internal form (gmsm_bases_register)."""

    def __init__(self, group, handle, n):
        self.group, self.handle, self.n = group, handle, n

    def precompute(self, c=0):
        """gmsm_bases_precompute: window tables 2^(c w) P_i in HBM (c = 0: the library's width); every later MultiExp over
        these bases fills one bucket set.  Returns the tables' window width."""
        L = _lib.load()
        rc = L.gmsm_bases_precompute(self.handle, int(c))
        if rc:
            raise RuntimeError(self.group._error(rc))
        return int(L.gmsm_bases_table_bits(self.handle))

    @property
    def table_bits(self):
        return int(_lib.load().gmsm_bases_table_bits(self.handle))

    def MultiExp(self, scalars, config=MultiExpConfig()):
        """MultiExp(bases[:len(scalars)], scalars): returns (jacobian_limbs, None) or (None, error)."""
        L = _lib.load()
        g = self.group
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, g.fr_limbs)
        out = np.zeros(g.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_bases(self.handle, _ptr(scalars), scalars.shape[0], int(config.NbTasks), _ptr(out))
        return (out, None) if rc == 0 else (None, g._error(rc))

    def MultiExpSharded(self, scalars, config=MultiExpConfig(), mode="auto"):
        """gmsm_multiexp_bases_sharded (handles of register_bases_sharded only): explicit decomposition."""
        L = _lib.load()
        g = self.group
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, g.fr_limbs)
        out = np.zeros(g.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_bases_sharded(self.handle, _ptr(scalars), scalars.shape[0], int(config.NbTasks),
                                           SHARD_MODES[mode], _ptr(out))
        return (out, None) if rc == 0 else (None, g._error(rc))

    def multiexp_device(self, d_scalars, n, stream=0):
        L = _lib.load()
        out = np.zeros(self.group.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_bases_device(self.handle, d_scalars, n, stream or None, _ptr(out))
        if rc:
            raise RuntimeError(self.group._error(rc))
        return out

    def MultiExpBatch(self, scalars=None, d_scalars=None, n=None, k=None, stream=0):
        """gmsm_multiexp_bases_batch: k MultiExp over bases[:n], one per scalar vector; scalars is a (k, n, fr_limbs) host
        array or d_scalars a device pointer to the same layout. Returns ((k, jac_limbs) array, None) or (None, error)."""
        L = _lib.load()
        g = self.group
        if scalars is not None:
            scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
            k, n = scalars.shape[0], scalars.shape[1]
            scalars = scalars.reshape(k, n, g.fr_limbs)
        out = np.zeros((k, g.jac_limbs), dtype=np.uint64)
        rc = L.gmsm_multiexp_bases_batch(self.handle, _ptr(scalars) if scalars is not None and scalars.size else None,
                                         d_scalars, n, k, stream or None, _ptr(out))
        return (out, None) if rc == 0 else (None, g._error(rc))

    def submit(self, d_scalars, n, stream=0):
        """gmsm_multiexp_bases_submit: launch MultiExp(bases[:n], d_scalars) without waiting; returns a ticket."""
        import ctypes
        L = _lib.load()
        ticket = ctypes.c_uint64(0)
        rc = L.gmsm_multiexp_bases_submit(self.handle, d_scalars, n, stream or None, ctypes.byref(ticket))
        if rc:
            raise RuntimeError(self.group._error(rc))
        return ticket.value

    def collect(self, ticket):
        """gmsm_multiexp_collect: wait for a submitted MultiExp and return its Jacobian limbs."""
        L = _lib.load()
        out = np.zeros(self.group.jac_limbs, dtype=np.uint64)
        rc = L.gmsm_multiexp_collect(ticket, _ptr(out))
        if rc:
            raise RuntimeError(self.group._error(rc))
        return out

    def release(self):
        if self.handle:
            _lib.load().gmsm_bases_release(self.handle)
            self.handle = 0


class G1Jac(_Group):
    group = "g1"

    def MultiExp(self, points, scalars, config=MultiExpConfig()):
        """(*G1Jac).MultiExp (multiexp.go:32): returns (jacobian_limbs, None) or (None, error)."""
        return self._multiexp_jac(points, scalars, config)


class G1Affine(_Group):
    group = "g1"

    def MultiExp(self, points, scalars, config=MultiExpConfig()):
        """(*G1Affine).MultiExp (multiexp.go:20): returns (affine_limbs, None) or (None, error)."""
        return self._multiexp_affine(points, scalars, config)


class G2Jac(_Group):
    group = "g2"

    def MultiExp(self, points, scalars, config=MultiExpConfig()):
        return self._multiexp_jac(points, scalars, config)


class G2Affine(_Group):
    group = "g2"

    def MultiExp(self, points, scalars, config=MultiExpConfig()):
        return self._multiexp_affine(points, scalars, config)
