"""Curve/field constants: re-derived from the moduli, and (where the reference tree exists) compared with the
constants gnark-crypto's generator emitted."""
import importlib
import os
import re

import pytest

curves = importlib.import_module("gnark-crypto_amd.curves")
REF = "/root/reference/ecc"
REF_DIR = {"bn254": "bn254", "bls12_381": "bls12-381", "bw6_761": "bw6-761"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(curves.CURVES))
def test_basic_shape(name):
    c = curves.CURVES[name]
    assert c.p % 4 == 3 and pow(2, c.p - 1, c.p) == 1 and pow(2, c.r - 1, c.r) == 1
    x, y = c.g1
    assert (y * y - x * x * x - c.b) % c.p == 0
    # no-carry Montgomery condition (field/generator/config/field_config.go:200-206): spare top bit in both fields
    for q, n in ((c.p, c.fp_limbs), (c.r, c.fr_limbs)):
        top = q >> (64 * (n - 1))
        assert top < (1 << 63) - 1
    assert {"bn254": (4, 4, 254), "bls12_381": (6, 4, 255), "bw6_761": (12, 6, 377)}[name] == (c.fp_limbs, c.fr_limbs, c.fr_bits)


def _parse_header(path):
    txt = open(path).read()
    arrays = {m.group(1): [int(x, 16) for x in re.findall(r"0x([0-9a-f]+)ULL", m.group(2))]
              for m in re.finditer(r"static const uint64_t (\w+)\[\d+\] = \{([^}]*)\};", txt)}
    defs = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define (\w+) 0x([0-9a-f]+)ULL", txt)}
    return arrays, defs


@pytest.mark.parametrize("hdr", ["gnark-crypto_amd/csrc/gmsm_params.h", "oracle/oracle_params.h"])
def test_generated_headers_match_definitions(hdr):
    arrays, defs = _parse_header(os.path.join(ROOT, hdr))
    val = lambda limbs: sum(l << (64 * i) for i, l in enumerate(limbs))
    for c in curves.CURVES.values():
        for fld, q, n in (("fp", c.p, c.fp_limbs), ("fr", c.r, c.fr_limbs)):
            pre = f"{c.name}_{fld}"
            R = 1 << (64 * n)
            assert val(arrays[pre + "_q"]) == q
            assert val(arrays[pre + "_one"]) == R % q
            assert val(arrays[pre + "_rsquare"]) == R * R % q
            assert (defs[pre.upper() + "_QINVNEG"] * q + 1) % (1 << 64) == 0
        n = c.fp_limbs
        gx = val(arrays[c.name + "_g1_gen"][:n]) * pow(c.fp_R, -1, c.p) % c.p
        assert gx == c.g1[0]


def test_params32_header_consistent():
    txt = open(os.path.join(ROOT, "gnark-crypto_amd/csrc/gmsm_params32.h")).read()
    for c in curves.CURVES.values():
        for fld, q, n in (("fp", c.p, c.fp_limbs), ("fr", c.r, c.fr_limbs)):
            block = txt[txt.index(f"struct {c.name}_{fld}_params"):]
            block = block[: block.index("\n};")]
            arr = lambda nm: sum(int(x, 16) << (32 * i) for i, x in enumerate(re.findall(r"0x([0-9a-f]+)u", re.search(nm + r"\[\d+\] = \{([^}]*)\}", block).group(1))))
            R = 1 << (64 * n)
            assert arr("Q") == q and arr("ONE") == R % q and arr("RSQ") == R * R % q
            qinv = int(re.search(r"QINV = 0x([0-9a-f]+)u", block).group(1), 16)
            assert (qinv * q + 1) % (1 << 32) == 0


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", list(curves.CURVES))
def test_constants_match_reference(name):
    c = curves.CURVES[name]
    for fld, q, n in (("fp", c.p, c.fp_limbs), ("fr", c.r, c.fr_limbs)):
        txt = open(os.path.join(REF, REF_DIR[name], fld, "element.go")).read()
        limbs = [int(re.search(rf"\bq{i}\s*(?:uint64)?\s*=\s*(\d+)", txt).group(1)) for i in range(n)]
        assert sum(l << (64 * i) for i, l in enumerate(limbs)) == q
        qinv = int(re.search(r"const qInvNeg\s*(?:uint64)?\s*=\s*(\d+)", txt).group(1))
        assert (qinv * q + 1) % (1 << 64) == 0
        assert qinv == (-pow(q, -1, 1 << 64)) % (1 << 64)
        assert int(re.search(r"Limbs\s*=\s*(\d+)", txt).group(1)) == n
        assert int(re.search(r"Bits\s*=\s*(\d+)", txt).group(1)) == q.bit_length()
        # rSquare: first limb is listed in the source
        m = re.search(r"var rSquare = Element\{\s*((?:\d+,\s*)+)\}", txt)
        rs = [int(x) for x in re.findall(r"\d+", m.group(1))]
        R = 1 << (64 * n)
        assert sum(l << (64 * i) for i, l in enumerate(rs)) == R * R % q
