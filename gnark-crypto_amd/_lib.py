"""ctypes binding of libgmsm.so (C ABI declared in include/gmsm.h).

The library is built in-tree by `make -C gnark-crypto_amd/csrc` (see __graft_entry__.build()).  Loading fails loudly
when it is missing; there is no Python/CPU fallback for any compute entry.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GMSM_LIB") or os.path.join(CSRC, "libgmsm.so")  # GMSM_LIB: A/B builds

GMSM_OK, GMSM_ERR_LEN, GMSM_ERR_CONFIG, GMSM_ERR_DEVICE, GMSM_ERR_ARG, GMSM_ERR_POINT = 0, 1, 2, 3, 4, 5

GROUP_IDS = {
    ("bn254", "g1"): 0, ("bn254", "g2"): 1,
    ("bls12_381", "g1"): 2, ("bls12_381", "g2"): 3,
    ("bw6_761", "g1"): 4, ("bw6_761", "g2"): 5,
}

# every symbol include/gmsm.h declares (tests/test_abi.py checks the built library exports all of them)
ABI_SYMBOLS = [
    "gmsm_bn254_g1_multiexp", "gmsm_bn254_g2_multiexp", "gmsm_bls12_381_g1_multiexp", "gmsm_bls12_381_g2_multiexp",
    "gmsm_bw6_761_g1_multiexp", "gmsm_bw6_761_g2_multiexp", "gmsm_multiexp", "gmsm_multiexp_affine", "gmsm_fold",
    "gmsm_multiexp_device", "gmsm_bases_register", "gmsm_bases_release", "gmsm_multiexp_bases",
    "gmsm_multiexp_bases_device", "gmsm_multiexp_bases_submit", "gmsm_multiexp_collect", "gmsm_multiexp_bases_batch", "gmsm_default_window_bits", "gmsm_default_plan", "gmsm_num_windows", "gmsm_window_sums_device",
    "gmsm_window_sums_enqueue", "gmsm_fold_window_sets", "gmsm_fold_windows", "gmsm_batch_scalar_mul", "gmsm_batch_scalar_mul_device",
    "gmsm_batch_jac_to_affine", "gmsm_jac_to_affine", "gmsm_affine_limbs", "gmsm_scalar_limbs", "gmsm_debug_decompose",
    "gmsm_debug_field_op", "gmsm_debug_group_op", "gmsm_debug_glv_split", "gmsm_generate_points", "gmsm_set_profiling", "gmsm_get_stage_times",
    "gmsm_get_stage_launches", "gmsm_points_from_raw", "gmsm_points_validate", "gmsm_bases_register_raw",
    "gmsm_points_from_compressed", "gmsm_points_compress", "gmsm_bases_register_compressed",
    "gmsm_bases_register_dump", "gmsm_fft_domain_new", "gmsm_fft_domain_release", "gmsm_fft_domain_info", "gmsm_fft",
    "gmsm_fft_bit_reverse",
    "gmsm_bases_precompute", "gmsm_bases_table_bits", "gmsm_debug_table_runs", "gmsm_debug_small_runs", "gmsm_multiexp_sharded", "gmsm_bases_register_sharded", "gmsm_multiexp_bases_sharded", "gmsm_set_devices",
    "gmsm_get_devices", "gmsm_set_option", "gmsm_get_option", "gmsm_trim", "gmsm_shutdown",
    "gmsm_device_count", "gmsm_set_device", "gmsm_last_error",
    "gmsm_version",
]

_lib = None


def build(jobs=None):
    """Compile libgmsm.so for gfx950 (hipcc cross-compiles without a GPU)."""
    jobs = jobs or os.cpu_count() or 4
    subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}", "libgmsm.so"])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C gnark-crypto_amd/csrc` (or __graft_entry__.build()); "
                           "there is no fallback implementation")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1. Two HSA runtimes in one process
    # cannot both own the GPU, so when torch is going to share the process (device tensors, torch.distributed/RCCL) it
    # must be loaded first; libgmsm.so then binds to the already-loaded runtime by SONAME.
    if os.environ.get("GMSM_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p
    for name in ABI_SYMBOLS[:6]:
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [u64p, sz, u64p, sz, ctypes.c_int, u64p]
    L.gmsm_multiexp.restype = ctypes.c_int
    L.gmsm_multiexp.argtypes = [ctypes.c_int, u64p, sz, u64p, sz, ctypes.c_int, u64p]
    L.gmsm_multiexp_affine.restype = ctypes.c_int
    L.gmsm_multiexp_affine.argtypes = [ctypes.c_int, u64p, sz, u64p, sz, ctypes.c_int, u64p]
    L.gmsm_fold.restype = ctypes.c_int
    L.gmsm_fold.argtypes = [ctypes.c_int, u64p, sz, u64p, ctypes.c_int, u64p]
    L.gmsm_multiexp_device.restype = ctypes.c_int
    L.gmsm_multiexp_device.argtypes = [ctypes.c_int, vp, vp, sz, vp, u64p]
    L.gmsm_bases_register.restype = ctypes.c_int
    L.gmsm_bases_register.argtypes = [ctypes.c_int, u64p, vp, sz, vp]
    L.gmsm_bases_release.restype = ctypes.c_int
    L.gmsm_bases_release.argtypes = [ctypes.c_uint64]
    L.gmsm_bases_precompute.restype = ctypes.c_int
    L.gmsm_bases_precompute.argtypes = [ctypes.c_uint64, ctypes.c_uint]
    L.gmsm_bases_table_bits.restype = ctypes.c_uint
    L.gmsm_bases_table_bits.argtypes = [ctypes.c_uint64]
    L.gmsm_debug_table_runs.restype = ctypes.c_ulong
    L.gmsm_debug_table_runs.argtypes = []
    L.gmsm_debug_small_runs.restype = ctypes.c_ulong
    L.gmsm_debug_small_runs.argtypes = []
    L.gmsm_multiexp_bases.restype = ctypes.c_int
    L.gmsm_multiexp_bases.argtypes = [ctypes.c_uint64, u64p, sz, ctypes.c_int, u64p]
    L.gmsm_multiexp_bases_device.restype = ctypes.c_int
    L.gmsm_multiexp_bases_device.argtypes = [ctypes.c_uint64, vp, sz, vp, u64p]
    L.gmsm_multiexp_bases_submit.restype = ctypes.c_int
    L.gmsm_multiexp_bases_submit.argtypes = [ctypes.c_uint64, vp, sz, vp, ctypes.POINTER(ctypes.c_uint64)]
    L.gmsm_multiexp_bases_batch.restype = ctypes.c_int
    L.gmsm_multiexp_bases_batch.argtypes = [ctypes.c_uint64, u64p, vp, sz, sz, vp, u64p]
    L.gmsm_multiexp_collect.restype = ctypes.c_int
    L.gmsm_multiexp_collect.argtypes = [ctypes.c_uint64, u64p]
    L.gmsm_default_window_bits.restype = ctypes.c_uint
    L.gmsm_default_window_bits.argtypes = [ctypes.c_int, sz]
    L.gmsm_default_plan.restype = ctypes.c_int
    L.gmsm_default_plan.argtypes = [ctypes.c_int, sz] + [ctypes.POINTER(ctypes.c_uint)] * 4
    L.gmsm_num_windows.restype = ctypes.c_uint
    L.gmsm_num_windows.argtypes = [ctypes.c_int, ctypes.c_uint]
    L.gmsm_window_sums_device.restype = ctypes.c_int
    L.gmsm_window_sums_device.argtypes = [ctypes.c_int, vp, vp, sz, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp, u64p]
    L.gmsm_window_sums_enqueue.restype = ctypes.c_int
    L.gmsm_window_sums_enqueue.argtypes = [ctypes.c_int, vp, ctypes.c_uint64, vp, sz, ctypes.c_uint, ctypes.c_uint,
                                           ctypes.c_uint, vp, vp]
    L.gmsm_fold_window_sets.restype = ctypes.c_int
    L.gmsm_fold_window_sets.argtypes = [ctypes.c_int, ctypes.c_uint, u64p, ctypes.c_uint, u64p]
    L.gmsm_batch_scalar_mul.restype = ctypes.c_int
    L.gmsm_batch_scalar_mul.argtypes = [ctypes.c_int, u64p, u64p, sz, u64p]
    L.gmsm_batch_scalar_mul_device.restype = ctypes.c_int
    L.gmsm_batch_scalar_mul_device.argtypes = [ctypes.c_int, u64p, vp, sz, vp, vp]
    L.gmsm_batch_jac_to_affine.restype = ctypes.c_int
    L.gmsm_batch_jac_to_affine.argtypes = [ctypes.c_int, u64p, sz, u64p]
    L.gmsm_fold_windows.restype = ctypes.c_int
    L.gmsm_fold_windows.argtypes = [ctypes.c_int, ctypes.c_uint, u64p, u64p]
    L.gmsm_jac_to_affine.restype = ctypes.c_int
    L.gmsm_jac_to_affine.argtypes = [ctypes.c_int, u64p, u64p]
    L.gmsm_affine_limbs.restype = sz
    L.gmsm_affine_limbs.argtypes = [ctypes.c_int]
    L.gmsm_scalar_limbs.restype = sz
    L.gmsm_scalar_limbs.argtypes = [ctypes.c_int]
    L.gmsm_debug_decompose.restype = ctypes.c_int
    L.gmsm_debug_decompose.argtypes = [ctypes.c_int, u64p, sz, ctypes.c_uint, vp]
    L.gmsm_debug_field_op.restype = ctypes.c_int
    L.gmsm_debug_field_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, u64p, sz, u64p]
    L.gmsm_debug_group_op.restype = ctypes.c_int
    L.gmsm_debug_group_op.argtypes = [ctypes.c_int, ctypes.c_int, u64p, u64p, sz, u64p]
    L.gmsm_generate_points.restype = ctypes.c_int
    L.gmsm_generate_points.argtypes = [ctypes.c_int, u64p, u64p, u64p, ctypes.c_int, sz, ctypes.c_int, u64p]
    L.gmsm_set_profiling.restype = None
    L.gmsm_set_profiling.argtypes = [ctypes.c_int]
    L.gmsm_get_stage_times.restype = ctypes.c_int
    L.gmsm_get_stage_times.argtypes = [vp, ctypes.c_int, vp]
    i64p = ctypes.POINTER(ctypes.c_int64)
    L.gmsm_points_from_raw.restype = ctypes.c_int
    L.gmsm_points_from_raw.argtypes = [ctypes.c_int, vp, sz, ctypes.c_int, u64p, vp, i64p]
    L.gmsm_points_validate.restype = ctypes.c_int
    L.gmsm_points_validate.argtypes = [ctypes.c_int, u64p, vp, sz, ctypes.c_int, i64p]
    L.gmsm_bases_register_raw.restype = ctypes.c_int
    L.gmsm_bases_register_raw.argtypes = [ctypes.c_int, vp, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), i64p]
    L.gmsm_points_from_compressed.restype = ctypes.c_int
    L.gmsm_points_from_compressed.argtypes = [ctypes.c_int, vp, sz, ctypes.c_int, u64p, vp, i64p]
    L.gmsm_points_compress.restype = ctypes.c_int
    L.gmsm_points_compress.argtypes = [ctypes.c_int, u64p, vp, sz, vp]
    L.gmsm_bases_register_compressed.restype = ctypes.c_int
    L.gmsm_bases_register_compressed.argtypes = [ctypes.c_int, vp, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), i64p]
    L.gmsm_bases_register_dump.restype = ctypes.c_int
    L.gmsm_bases_register_dump.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, sz, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(sz), i64p]
    L.gmsm_fft_domain_new.restype = ctypes.c_int
    L.gmsm_fft_domain_new.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.gmsm_fft_domain_release.restype = ctypes.c_int
    L.gmsm_fft_domain_release.argtypes = [ctypes.c_uint64]
    L.gmsm_fft_domain_info.restype = ctypes.c_int
    L.gmsm_fft_domain_info.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), u64p, u64p, u64p, u64p, u64p]
    L.gmsm_fft.restype = ctypes.c_int
    L.gmsm_fft.argtypes = [ctypes.c_uint64, u64p, vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.gmsm_fft_bit_reverse.restype = ctypes.c_int
    L.gmsm_fft_bit_reverse.argtypes = [ctypes.c_int, u64p, vp, sz, vp]
    L.gmsm_get_stage_launches.restype = ctypes.c_int
    L.gmsm_get_stage_launches.argtypes = [vp, ctypes.c_int]
    ip = ctypes.POINTER(ctypes.c_int)
    L.gmsm_multiexp_sharded.restype = ctypes.c_int
    L.gmsm_multiexp_sharded.argtypes = [ctypes.c_int, u64p, sz, u64p, sz, ctypes.c_int, ip, ctypes.c_int, ctypes.c_int, u64p]
    L.gmsm_bases_register_sharded.restype = ctypes.c_int
    L.gmsm_bases_register_sharded.argtypes = [ctypes.c_int, u64p, sz, ip, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    L.gmsm_multiexp_bases_sharded.restype = ctypes.c_int
    L.gmsm_multiexp_bases_sharded.argtypes = [ctypes.c_uint64, u64p, sz, ctypes.c_int, ctypes.c_int, u64p]
    L.gmsm_set_devices.restype = ctypes.c_int
    L.gmsm_set_devices.argtypes = [ip, ctypes.c_int]
    L.gmsm_get_devices.restype = ctypes.c_int
    L.gmsm_get_devices.argtypes = [ip, ctypes.c_int]
    L.gmsm_debug_glv_split.restype = ctypes.c_int
    L.gmsm_debug_glv_split.argtypes = [ctypes.c_int, u64p, sz, vp]
    L.gmsm_set_option.restype = ctypes.c_int
    L.gmsm_set_option.argtypes = [ctypes.c_int, ctypes.c_uint]
    L.gmsm_get_option.restype = ctypes.c_uint
    L.gmsm_get_option.argtypes = [ctypes.c_int]
    L.gmsm_trim.restype = ctypes.c_int
    L.gmsm_trim.argtypes = [sz, ctypes.POINTER(sz)]
    L.gmsm_shutdown.restype = ctypes.c_int
    L.gmsm_shutdown.argtypes = []
    L.gmsm_device_count.restype = ctypes.c_int
    L.gmsm_set_device.restype = ctypes.c_int
    L.gmsm_set_device.argtypes = [ctypes.c_int]
    L.gmsm_last_error.restype = ctypes.c_char_p
    L.gmsm_version.restype = ctypes.c_char_p
    _lib = L
    return L


def effective_cpus():
    """Cores this process may use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def last_error():
    return load().gmsm_last_error().decode()


# enum gmsm_option (include/gmsm.h)
OPTIONS = {"window_bits": 0, "tables": 1, "max_run": 2, "host_ranges": 3, "fixed_base_bits": 4, "spin_wait_us": 5, "small_bits": 6, "small_max": 7, "split": 8, "glv": 9, "small_quad": 10}


def set_option(name, value):
    rc = load().gmsm_set_option(OPTIONS[name], int(value))
    if rc:
        raise ValueError("gmsm: " + last_error())


def get_option(name):
    return int(load().gmsm_get_option(OPTIONS[name]))


class options:
    """with options(window_bits=11, max_run=4096): ...  - process-wide switches of the library (gmsm_set_option), restored
    on exit.  The tests reach the point-range splits and the forced widths through this, not through the environment."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def trim(keep_bytes=0):
    """gmsm_trim: scratch of the idle workspaces back to the device; returns the bytes released."""
    freed = ctypes.c_size_t(0)
    rc = load().gmsm_trim(keep_bytes, ctypes.byref(freed))
    if rc:
        raise RuntimeError("gmsm: " + last_error())
    return int(freed.value)


def shutdown():
    rc = load().gmsm_shutdown()
    if rc:
        raise RuntimeError("gmsm: " + last_error())
