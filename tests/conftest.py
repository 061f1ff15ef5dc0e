import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def gm():
    """The product package (host mirror + ctypes binding of libgmsm.so)."""
    return importlib.import_module("gnark-crypto_amd")


@pytest.fixture
def forced_options(gm):
    """forced_options(max_run=4096, ...): library switches (gmsm_set_option) for the rest of the test, restored after it."""
    old = {}

    def force(**kw):
        for k, v in kw.items():
            old.setdefault(k, gm.get_option(k))
            gm.set_option(k, v)
    yield force
    for k, v in old.items():
        gm.set_option(k, v)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # noqa: E402  (oracle/oracle.py, test infrastructure)
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pyref_mod():
    import pyref  # noqa: E402
    return pyref


ALL_GROUPS = [("bn254", "g1"), ("bn254", "g2"), ("bls12_381", "g1"), ("bls12_381", "g2"), ("bw6_761", "g1"), ("bw6_761", "g2")]

SEED = 0x6D736D


def rng_for(*tags):
    return np.random.default_rng([SEED] + [int(t) & 0xFFFFFFFF for t in tags])


def random_field_limbs(rng, modulus, nlimbs, count):
    """count values uniform in [0, modulus) as little-endian uint64 limbs (rejection sampling on the whole value)."""
    out = np.zeros((count, nlimbs), dtype=np.uint64)
    top_bits = modulus.bit_length() - 64 * (nlimbs - 1)
    mod_limbs = [(modulus >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs)]
    todo = np.arange(count)
    while todo.size:
        cand = rng.integers(0, 2**64, size=(todo.size, nlimbs), dtype=np.uint64)
        if top_bits < 64:
            cand[:, -1] &= np.uint64((1 << top_bits) - 1)
        # lexicographic compare from the top limb
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for i in range(nlimbs - 1, -1, -1):
            m = np.uint64(mod_limbs[i])
            lt |= eq & (cand[:, i] < m)
            eq &= cand[:, i] == m
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out


def random_scalars(rng, curve, count):
    """Uniform scalars: the stored (Montgomery) limbs are uniform in [0, r), hence so is the scalar value."""
    return random_field_limbs(rng, curve.r, curve.fr_limbs, count)


def int_to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def scalars_from_ints(curve, values):
    R = curve.fr_R
    return np.array([int_to_limbs(v % curve.r * R % curve.r, curve.fr_limbs) for v in values], dtype=np.uint64).reshape(-1, curve.fr_limbs)
