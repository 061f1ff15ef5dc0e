"""The reference's COMPRESSED point encoding restated with Python integers - test infrastructure, like oracle/pyref.py.
Follows (*G1Affine).Bytes / setBytes (ecc/bn254/marshal.go:801-823, :862-948; G2 :1051-1075, :1118-1215), the 3-bit flags of
BLS12-381 / BW6-761 (ecc/bls12-381/marshal.go:25-35), LexicographicallyLargest (fp/element.go:282-296, E2:
internal/fptower/e2.go:46-52) and, for the square root over Fp2, the reference's own algorithm (E2.Sqrt, e2.go:211-234, in
subgroup_points.sqrt_fp2) - NOT the method the device uses (gmsm_decompress.h: norm + two Fp roots), which makes the GPU
comparison a cross-check of two algorithms."""
from subgroup_points import curve_b, sqrt_fp, sqrt_fp2

ERR_SQRT = "invalid compressed coordinate: square root doesn't exist"
ERR_INFINITY = "invalid infinity point encoding"
ERR_ELEMENT = "invalid fp.Element encoding"
ERR_FLAG = "invalid point encoding"


def flags(curve_name):
    """(flag bits, smallest, largest, infinity) - marshal.go:26-30 / bls12-381/marshal.go:25-35"""
    return (2, 0b10, 0b11, 0b01) if curve_name == "bn254" else (3, 0b100, 0b101, 0b110)


def lex_largest(pg, y):
    half = (pg.p - 1) // 2
    if pg.ext == 1:
        return y > half
    return (y.a1 > half) if y.a1 != 0 else (y.a0 > half)


def neg(pg, y):
    return (-y) % pg.p if pg.ext == 1 else -y


def compressed_size(pg):
    return 8 * pg.c.fp_limbs * pg.ext


def encode_compressed(pg, P):
    """Bytes(): the infinity flag over zeroes, or X (Fp2: A1 | A0) with the flag of the half Y lies in."""
    bits, small, large, inf = flags(pg.c.name)
    nb = 8 * pg.c.fp_limbs
    if P is None:
        out = bytearray(compressed_size(pg))
        out[0] = inf << (8 - bits)
        return bytes(out)
    x, y = P
    out = bytearray(x.to_bytes(nb, "big") if pg.ext == 1 else x.a1.to_bytes(nb, "big") + x.a0.to_bytes(nb, "big"))
    out[0] |= (large if lex_largest(pg, y) else small) << (8 - bits)
    return bytes(out)


def decode_compressed(pyref, pg, buf):
    """setBytes, compressed branch: returns the point (None = infinity) or raises ValueError with the reference's text."""
    bits, small, large, inf = flags(pg.c.name)
    nb = 8 * pg.c.fp_limbs
    assert len(buf) == compressed_size(pg)
    flag = buf[0] >> (8 - bits)
    body = bytes([buf[0] & (0xff >> bits)]) + bytes(buf[1:])
    if flag == inf:
        if any(body):
            raise ValueError(ERR_INFINITY)
        return None
    if flag not in (small, large):
        raise ValueError(ERR_FLAG)
    vals = [int.from_bytes(body[k * nb:(k + 1) * nb], "big") for k in range(pg.ext)]
    if any(v >= pg.p for v in vals):
        raise ValueError(ERR_ELEMENT)
    if pg.ext == 1:
        x = vals[0]
        y = sqrt_fp((x * x * x + curve_b(pyref, pg)) % pg.p, pg.p)
    else:
        x = pyref.Fp2(vals[1], vals[0], pg.p)
        y = sqrt_fp2(pyref, x * x * x + curve_b(pyref, pg))
    if y is None:
        raise ValueError(ERR_SQRT)
    if lex_largest(pg, y) != (flag == large):
        y = neg(pg, y)
    return (x, y)


def pow_q4_schedule(x, q):
    """The device's x^(q >> 2) (gmsm_decompress.h::fpu_pow_q4) step for step: left to right, a sliding window of three bits over
    x, x^3, x^5, x^7, one squaring per exponent bit below the first window. Returns (value, squarings, products)."""
    bit = lambda j: (q >> (j + 2)) & 1
    x2 = x * x % q
    odd = {1: x, 3: x * x2 % q}
    odd[5] = odd[3] * x2 % q
    odd[7] = odd[5] * x2 % q
    sq, mu = 1, 3
    r, started, mulpos, val = x, False, -1, 0
    for j in range(q.bit_length() - 3, -1, -1):
        if mulpos < 0 and bit(j):
            mulpos = j - 2 if j >= 2 else 0
            while not bit(mulpos):
                mulpos += 1
            val = 0
            for k in range(j, mulpos - 1, -1):
                val = (val << 1) | bit(k)
        if started:
            r, sq = r * r % q, sq + 1
        if j == mulpos:
            if started:
                r, mu = r * odd[val] % q, mu + 1
            else:
                r = odd[val]
            started, mulpos = True, -1
    return r, sq, mu
