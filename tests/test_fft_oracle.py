"""Pins the fr/fft oracle (oracle/fft_tmpl.h) with the reference's own test properties (ecc/bn254/fr/fft/fft_test.go:22-200),
evaluated with Python big integers - an implementation that shares nothing with the C code:
  * DIF FFT, then BitReverse: entry i equals the polynomial evaluated at Generator^i (on cosets: at u * Generator^i);
  * DIT FFT of the bit-reversed input gives the same evaluations;
  * bitReverse(FFTInverse_DIF(FFT_DIT(bitReverse(p)))) == p, also on cosets; FFT_DIT(FFTInverse_DIF(p)) == p;
  * fr.Generator(m) has exact order NextPowerOfTwo(m) (fr/generator.go:18-36)."""
import numpy as np
import pytest

from conftest import random_field_limbs, rng_for

CURVE_NAMES = ["bn254", "bls12_381", "bw6_761"]


def to_int(limbs):
    return sum(int(v) << (64 * i) for i, v in enumerate(limbs))


def from_mont(c, limbs):
    return to_int(limbs) * pow(c.fr_R, -1, c.r) % c.r


def to_mont_arr(c, vals):
    return np.array([[(v * c.fr_R % c.r >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(c.fr_limbs)] for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("curve", CURVE_NAMES)
def test_generator_order(oracle_mod, curve):
    F = oracle_mod.FFT(curve)
    c = F.curve
    for m in (1, 2, 3, 8, 1000, 1 << 20):
        x = 1
        while x < m:
            x <<= 1
        g = from_mont(c, F.generator(m))
        assert pow(g, x, c.r) == 1 and (x == 1 or pow(g, x // 2, c.r) == c.r - 1)
        assert g == pow(c.fr_root_of_unity, 1 << (c.fr_max_order - x.bit_length() + 1), c.r)
    assert F.generator(1 << (c.fr_max_order + 1)) is None  # "the required root of unity does not exist"


@pytest.mark.parametrize("curve", CURVE_NAMES)
@pytest.mark.parametrize("logn", [1, 2, 3, 6, 8])
def test_fft_equals_polynomial_evaluation(oracle_mod, curve, logn):
    F = oracle_mod.FFT(curve)
    c = F.curve
    n = 1 << logn
    rng = rng_for(71, logn, c.fr_limbs)
    pol = random_field_limbs(rng, c.r, c.fr_limbs, n)
    coeffs = [from_mont(c, p) for p in pol]
    w = from_mont(c, F.generator(n))
    u = c.fr_mult_gen

    def evaluate(x):
        acc = 0
        for co in reversed(coeffs):
            acc = (acc * x + co) % c.r
        return acc

    want = to_mont_arr(c, [evaluate(pow(w, i, c.r)) for i in range(n)])
    want_coset = to_mont_arr(c, [evaluate(u * pow(w, i, c.r) % c.r) for i in range(n)])
    dif = F.bit_reverse(F.transform(pol, decimation=oracle_mod.DIF))
    assert (dif == want).all()
    dit = F.transform(F.bit_reverse(pol), decimation=oracle_mod.DIT)
    assert (dit == want).all()
    assert (F.bit_reverse(F.transform(pol, decimation=oracle_mod.DIF, coset=True)) == want_coset).all()
    assert (F.transform(F.bit_reverse(pol), decimation=oracle_mod.DIT, coset=True) == want_coset).all()


@pytest.mark.parametrize("curve", CURVE_NAMES)
def test_fft_round_trips(oracle_mod, curve):
    F = oracle_mod.FFT(curve)
    c = F.curve
    for n in (1, 2, 16, 1024):
        pol = random_field_limbs(rng_for(72, n), c.r, c.fr_limbs, n)
        for coset in (False, True):
            x = F.transform(F.bit_reverse(pol), decimation=oracle_mod.DIT, coset=coset)
            x = F.bit_reverse(F.transform(x, inverse=True, decimation=oracle_mod.DIF, coset=coset))
            assert (x == pol).all()
            y = F.transform(F.transform(pol, inverse=True, decimation=oracle_mod.DIF, coset=coset), decimation=oracle_mod.DIT, coset=coset)
            assert (y == pol).all()
    with pytest.raises(ValueError):
        F.transform(np.zeros((3, c.fr_limbs), dtype=np.uint64))
