"""fr/fft on the device (gmsm_fft_*, gnark-crypto_amd/fft.py) against the oracle, bit for bit: every decimation,
direction and coset combination, sizes 1 .. 2^16 for the three scalar fields, 2^22 for BN254, plus the reference's
round-trip properties (ecc/bn254/fr/fft/fft_test.go) at a size the oracle does not need to touch."""
import numpy as np
import pytest

from conftest import random_field_limbs, rng_for

pytestmark = pytest.mark.gpu

CURVE_NAMES = ["bn254", "bls12_381", "bw6_761"]


@pytest.mark.parametrize("curve", CURVE_NAMES)
def test_domain_constants(gm, oracle_mod, curve):
    F = oracle_mod.FFT(curve)
    c = F.curve
    fr = oracle_mod.Field(f"{c.name}_fr", c.fr_limbs)
    for m in (1, 5, 1 << 10, (1 << 16) + 1):
        d = gm.fft.NewDomain(curve, m)
        x = 1
        while x < m:
            x <<= 1
        assert d.Cardinality == x
        assert (d.Generator == F.generator(m)).all()
        assert (fr.mul(d.Generator, d.GeneratorInv) == fr.to_mont(np.array([1] + [0] * (c.fr_limbs - 1), dtype=np.uint64))).all()
        one = fr.to_mont(np.array([1] + [0] * (c.fr_limbs - 1), dtype=np.uint64))
        assert (fr.mul(d.FrMultiplicativeGen, d.FrMultiplicativeGenInv) == one).all()
        card = fr.to_mont(np.array([x] + [0] * (c.fr_limbs - 1), dtype=np.uint64))
        assert (fr.mul(card, d.CardinalityInv) == one).all()
        d.release()
    with pytest.raises(ValueError):
        gm.fft.NewDomain(curve, 1 << (c.fr_max_order + 1))


@pytest.mark.parametrize("curve", CURVE_NAMES)
@pytest.mark.parametrize("logn", [0, 1, 2, 5, 10, 16])
def test_fft_matches_oracle(gm, oracle_mod, curve, logn):
    F = oracle_mod.FFT(curve)
    c = F.curve
    n = 1 << logn
    a = random_field_limbs(rng_for(81, logn, c.fr_limbs), c.r, c.fr_limbs, n)
    d = gm.fft.NewDomain(curve, n)
    for inverse in (False, True):
        for dec in (gm.fft.DIT, gm.fft.DIF):
            for coset in (False, True):
                opts = (gm.fft.OnCoset(),) if coset else ()
                got = (d.FFTInverse if inverse else d.FFT)(a, dec, *opts)
                want = F.transform(a, inverse=inverse, decimation=dec, coset=coset)
                assert (got == want).all(), (inverse, dec, coset)
    assert (gm.fft.BitReverse(curve, a) == F.bit_reverse(a)).all()
    d.release()


def test_fft_errors(gm):
    d = gm.fft.NewDomain("bn254", 64)
    with pytest.raises(ValueError):
        d.FFT(np.zeros((32, 4), dtype=np.uint64), gm.fft.DIF)  # len(a) != cardinality
    with pytest.raises(ValueError):
        gm.fft.BitReverse("bn254", np.zeros((24, 4), dtype=np.uint64))  # "len(a) must be a power of 2"
    d.release()


def test_fft_2_pow_22_device_resident(gm, oracle_mod):
    """BN254, 2^22 coefficients resident in HBM: DIF then the inverse DIT restores the input (fft_test.go:160-180), the
    forward result equals the oracle's, and the coset pair does the same."""
    import torch
    curve = "bn254"
    F = oracle_mod.FFT(curve)
    c = F.curve
    n = 1 << 22
    a = random_field_limbs(rng_for(82, 22), c.r, c.fr_limbs, n)
    d = gm.fft.NewDomain(curve, n)
    t = torch.from_numpy(a.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    d.fft_device(t.data_ptr(), gm.fft.DIF, stream=stream)
    got = t.cpu().numpy().view(np.uint64)
    assert (got == F.transform(a, decimation=oracle_mod.DIF)).all()
    d.fft_device(t.data_ptr(), gm.fft.DIT, inverse=True, stream=stream)
    assert (t.cpu().numpy().view(np.uint64) == a).all()
    d.fft_device(t.data_ptr(), gm.fft.DIF, gm.fft.OnCoset(), stream=stream)
    d.fft_device(t.data_ptr(), gm.fft.DIT, gm.fft.OnCoset(), inverse=True, stream=stream)
    assert (t.cpu().numpy().view(np.uint64) == a).all()
    gm.fft.BitReverse(curve, d_a=t.data_ptr(), n=n, stream=stream)
    gm.fft.BitReverse(curve, d_a=t.data_ptr(), n=n, stream=stream)
    assert (t.cpu().numpy().view(np.uint64) == a).all()
    d.release()
