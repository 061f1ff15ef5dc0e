"""MI355X-native multi-scalar multiplication for gnark-crypto's ecc/<curve>.MultiExp path.

Import with importlib (the directory name carries a hyphen):
    gm = importlib.import_module("gnark-crypto_amd")
"""
from .curves import BLS12_381, BN254, BW6_761, CURVES  # noqa: F401
from .multiexp import G1Affine, G1Jac, G2Affine, G2Jac, MultiExpConfig, get_devices, set_devices  # noqa: F401
from ._lib import options, set_option, get_option, trim, shutdown  # noqa: F401  (gmsm_set_option / gmsm_trim / gmsm_shutdown)
from . import fft  # noqa: F401  (fr/fft mirror: fft.NewDomain, fft.DIT / fft.DIF, fft.OnCoset, fft.BitReverse)
