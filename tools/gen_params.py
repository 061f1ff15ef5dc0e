#!/usr/bin/env python3
"""Generate the per-curve constant headers from gnark-crypto_amd/curves.py.

Writes two files with the same numbers but different consumers (the product must never include
anything under oracle/, and the oracle must stand alone):

  gnark-crypto_amd/csrc/gmsm_params.h   -- HIP kernels + C-ABI host code
  oracle/oracle_params.h                -- CPU oracle

Montgomery constants follow the reference's definitions (ecc/bn254/fp/element.go:44-74):
  q           modulus, little-endian 64-bit limbs
  qInvNeg     -q^-1 mod 2^64
  one         R mod q           (R = 2^(64*limbs))
  rSquare     R^2 mod q
tests/test_params.py checks them against the reference's generated constants when the tree is present.
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
curves = importlib.import_module("gnark-crypto_amd.curves")


def limbs64(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def carr(name, vals, ctype="uint64_t", static="static const"):
    body = ", ".join(f"0x{v:016x}ULL" for v in vals)
    return f"{static} {ctype} {name}[{len(vals)}] = {{{body}}};\n"


def field_block(prefix, q, n):
    R = 1 << (64 * n)
    out = f"#define {prefix.upper()}_LIMBS {n}\n"
    out += f"#define {prefix.upper()}_BITS {q.bit_length()}\n"
    out += carr(f"{prefix}_q", limbs64(q, n))
    out += f"#define {prefix.upper()}_QINVNEG 0x{(-pow(q, -1, 1 << 64)) % (1 << 64):016x}ULL\n"
    out += carr(f"{prefix}_one", limbs64(R % q, n))
    out += carr(f"{prefix}_rsquare", limbs64(R * R % q, n))
    return out


def curve_block(c):
    nfp, nfr = c.fp_limbs, c.fr_limbs
    R = c.fp_R
    out = f"/* ---- {c.name} ---- */\n"
    out += field_block(f"{c.name}_fp", c.p, nfp)
    out += field_block(f"{c.name}_fr", c.r, nfr)
    mont = lambda v: limbs64(v * R % c.p, nfp)
    Rr = c.fr_R
    out += carr(f"{c.name}_fr_root_of_unity", limbs64(c.fr_root_of_unity * Rr % c.r, nfr))
    out += f"#define {c.name.upper()}_FR_MAX_ORDER {c.fr_max_order}\n"
    out += carr(f"{c.name}_fr_mult_gen", limbs64(c.fr_mult_gen * Rr % c.r, nfr))
    out += carr(f"{c.name}_g1_gen", mont(c.g1[0]) + mont(c.g1[1]))
    if c.g2_ext == 2:
        (x0, x1), (y0, y1) = c.g2
        out += carr(f"{c.name}_g2_gen", mont(x0) + mont(x1) + mont(y0) + mont(y1))
    else:
        out += carr(f"{c.name}_g2_gen", mont(c.g2[0]) + mont(c.g2[1]))
    out += f"#define {c.name.upper()}_G2_EXT {c.g2_ext}\n\n"
    return out


def limbs32(v, n64):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(2 * n64)]


def glv_block(glv):
    """Scalar-decomposition constants of the curve (curves.GlvParams), 32-bit words; the A_ij as two's complement mod 2^(32 HL)."""
    w = lambda v, n: [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    arr = lambda name, vals: f"    static constexpr uint32_t {name}[{len(vals)}] = {{" + ", ".join(f"0x{x:08x}u" for x in vals) + "};\n"
    mod = 1 << (32 * glv.hl)
    out = "    /* GLV split s = k1 + k2 lambda (ecc/utils.go:62-170; derivation and bounds: gnark-crypto_amd/curves.py GlvParams) */\n"
    out += f"    static constexpr int GLV_BITS = {glv.bits};  /* |k1|, |k2| < 2^GLV_BITS */\n"
    out += f"    static constexpr int GLV_HL = {glv.hl};    /* words of a half scalar (two's complement) */\n"
    out += f"    static constexpr int GLV_SH = {glv.sh};    /* m_i = (s B_i) >> 32 GLV_SH */\n"
    out += f"    static constexpr int GLV_NB = {glv.nb};\n"
    out += arr("GLV_B1", w(glv.b1, glv.nb)) + arr("GLV_B2", w(glv.b2, glv.nb))
    for name, v in zip(("GLV_A11", "GLV_A21", "GLV_A12", "GLV_A22"), glv.a):
        out += arr(name, w(v % mod, glv.hl))
    return out


def cxx_field(prefix, q, n64, fft=None, glv=None):
    """constexpr 32-bit-limb parameter struct for the HIP kernels (folded into instruction literals).
    fft = (root_of_unity, max_order, mult_gen): the scalar field's FFT constants (fr/generator.go, fr/fft/domain.go)."""
    R = 1 << (64 * n64)
    n = 2 * n64
    arr = lambda name, v: f"    static constexpr uint32_t {name}[{n}] = {{" + ", ".join(f"0x{x:08x}u" for x in limbs32(v, n64)) + "};\n"
    out = f"struct {prefix}_params {{\n"
    out += f"    static constexpr int N = {n};           /* 32-bit limbs */\n"
    out += f"    static constexpr int BITS = {q.bit_length()};\n"
    out += arr("Q", q)
    out += f"    static constexpr uint32_t QINV = 0x{(-pow(q, -1, 1 << 32)) % (1 << 32):08x}u; /* -q^-1 mod 2^32 */\n"
    out += arr("ONE", R % q)
    out += arr("RSQ", R * R % q)
    if q % 4 == 3:  # the base fields (sqrt = x^((q+1)/4); point decompression, gmsm_decompress.h)
        out += arr("INV2", (q + 1) // 2 * R % q)   # 1/2, Montgomery
        out += arr("LEX_HALF", (q + 1) // 2)       # regular form: z >= (q+1)/2  <=>  LexicographicallyLargest (fp/element.go:282-296)
    out += unsat_block(q, n64)
    if fft:
        root, max_order, gen = fft
        assert pow(root, 1 << max_order, q) == 1 and pow(root, 1 << (max_order - 1), q) == q - 1
        out += "    /* FFT domain constants, Montgomery form: primitive 2^MAX_ORDER-th root of unity, generator of Fr^* (coset shift) */\n"
        out += arr("ROOT_OF_UNITY", root * R % q)
        out += f"    static constexpr unsigned MAX_ORDER = {max_order};\n"
        out += arr("MULT_GEN", gen * R % q)
    if glv is not None:
        out += glv_block(glv)
    out += "};\n"
    return out


# Unsaturated ("lazy") representation used by the bucket-accumulation hot loop: UL limbs of UW bits, Montgomery
# radix 2^(UL*UW).  UW is chosen so that a column of 2*UL products of (UW+1)-bit limbs fits a 64-bit accumulator.
UNSAT = {254: (9, 29), 381: (14, 28), 761: (28, 28), 255: (9, 29), 377: (14, 28)}


def redundant_multiple(q, k, UL, UW, borrow_units=4):
    """k*q written with limbs K_i = k_i + borrow_units*2^UW (i < UL-1), top limb reduced accordingly, so that
    a_i + K_i - b_i never underflows for b_i < borrow_units*2^UW.  sum K_i 2^(UW i) == k*q exactly."""
    v = k * q
    limbs = [(v >> (UW * i)) & ((1 << UW) - 1) for i in range(UL - 1)] + [v >> (UW * (UL - 1))]
    for i in range(UL - 1):
        limbs[i] += borrow_units << UW
        limbs[i + 1] -= borrow_units
    assert limbs[-1] > 0 and sum(l << (UW * i) for i, l in enumerate(limbs)) == v
    assert all(l < (1 << 32) for l in limbs)
    return limbs


def unsat_block(q, n64):
    UL, UW = UNSAT[q.bit_length()]
    T = UL * UW
    S = 64 * n64
    mask = (1 << UW) - 1
    ulimbs = lambda v: [(v >> (UW * i)) & mask for i in range(UL - 1)] + [v >> (UW * (UL - 1))]
    uarr = lambda name, vals: f"    static constexpr uint32_t {name}[{UL}] = {{" + ", ".join(f"0x{x:08x}u" for x in vals) + "};\n"
    out = f"    /* unsaturated form: {UL} limbs x {UW} bits, Montgomery radix 2^{T} */\n"
    out += f"    static constexpr int UL = {UL};\n    static constexpr int UW = {UW};\n"
    out += uarr("UQ", ulimbs(q))
    out += f"    static constexpr uint32_t UQINV = 0x{(-pow(q, -1, 1 << UW)) % (1 << UW):08x}u; /* -q^-1 mod 2^UW */\n"
    out += uarr("UONE", ulimbs((1 << T) % q))                      # 1 in the 2^T Montgomery domain
    out += uarr("UCIN", ulimbs(pow(2, 2 * T - S, q)))              # montmul'(x*2^S, UCIN) = x*2^T
    out += uarr("UCOUT", ulimbs(pow(2, S, q)))                     # montmul'(x*2^T, UCOUT) = x*2^S
    out += uarr("UK4", redundant_multiple(q, 4, UL, UW))
    out += uarr("UK4N", redundant_multiple(q, 4, UL, UW, borrow_units=1))  # 4q with limbs < 2^(UW+1): negation without a carry pass
    out += uarr("UK8", redundant_multiple(q, 8, UL, UW))
    out += uarr("UK8N", redundant_multiple(q, 8, UL, UW, borrow_units=1))  # 8q with limbs < 2^(UW+1): 8q - b for any normalised b < 4q, no carry pass
    out += uarr("UK16", redundant_multiple(q, 16, UL, UW))
    out += uarr("UK32", redundant_multiple(q, 32, UL, UW))         # bound-tracked Fp2 additions (operands up to 18q)
    out += uarr("UQ1", ulimbs(q)) + uarr("UQ2", ulimbs(2 * q))     # candidates for the exact zero test of a value < 3q
    out += uarr("UQ3", ulimbs(3 * q)) + uarr("UQ4", ulimbs(4 * q))  # reduced-class ([0,4q)) arithmetic of the Fp2 path
    return out


def fp2_inv(a0, a1, p):
    """(a0 + a1 u)^-1 in Fp[u]/(u^2+1)."""
    t = pow(a0 * a0 + a1 * a1, -1, p)
    return a0 * t % p, (-a1 * t) % p


def group_consts(c):
    """Curve coefficient b of y^2 = x^3 + b for G1 and for the twist G2 lives on, Montgomery form, 32-bit words
    (Fp2: a0 then a1), and the flag bits of the point encoding (ecc/<curve>/marshal.go, const block at the top).
    Twists: BN254 3/(9+u) (D-type, bn254.go:105-109), BLS12-381 4(1+u) (M-type, bls12-381.go:100-104), BW6-761 G2 over Fp:
    y^2 = x^3 + 4 (bw6-761.go:93-96)."""
    R = c.fp_R
    n64 = c.fp_limbs
    words = lambda v: limbs32(v % c.p * R % c.p, n64)
    b1 = words(c.b)
    if c.name == "bn254":
        i0, i1 = fp2_inv(9, 1, c.p)
        b2 = words(3 * i0) + words(3 * i1)
        flag_bits, inf_flag = 2, -1      # infinity = every byte zero under the "uncompressed" flag (marshal.go:826-834)
    elif c.name == "bls12_381":
        b2 = words(4) + words(4)
        flag_bits, inf_flag = 3, 0b010   # mUncompressedInfinity (bls12-381/marshal.go:27-34)
    else:
        b2 = words(4)
        flag_bits, inf_flag = 3, 0b010   # bw6-761/marshal.go:25-35
    # IsInSubGroup by endomorphism (gmsm_subgroup.h): which identity the reference tests, and its constants
    #   0 prime order (BN254 G1, g1.go:475-482)        1 [x^2] phi(P) + P = 0 (bls12-381/g1.go:481-492)
    #   2 [x] P + psi(P) = 0 (bls12-381/g2.go:484-491)   3 BN254 G2 (bn254/g2.go:483-497)   4 BW6-761 G1 / G2 (bw6-761/g1.go:482-496)
    w1 = c.third_root_one_g1 % c.p
    assert pow(w1, 3, c.p) == 1 and w1 != 1
    kinds = {"bn254": (0, 3), "bls12_381": (1, 2), "bw6_761": (4, 4)}[c.name]
    out = ""
    for gname, b, kind in (("g1", b1, kinds[0]), ("g2", b2, kinds[1])):
        out += f"struct {c.name}_{gname}_consts {{\n"
        out += f"    static constexpr uint32_t B[{len(b)}] = {{" + ", ".join(f"0x{x:08x}u" for x in b) + "};  /* curve coefficient, Montgomery */\n"
        out += f"    static constexpr int RAW_FLAG_BITS = {flag_bits};      /* metadata bits on top of the first byte */\n"
        out += f"    static constexpr int RAW_INFINITY_FLAG = {inf_flag};  /* flag value of an uncompressed point at infinity; -1: none */\n"
        comp = (0b10, 0b11, 0b01) if flag_bits == 2 else (0b100, 0b101, 0b110)   # marshal.go:26-30 / bls12-381/marshal.go:25-35
        out += f"    static constexpr int COMP_SMALLEST = {comp[0]}, COMP_LARGEST = {comp[1]}, COMP_INFINITY = {comp[2]};  /* compressed encodings: flag values */\n"
        out += f"    static constexpr int SUBGROUP_TEST = {kind};   /* which endomorphism identity IsInSubGroup tests (gmsm_subgroup.h) */\n"
        out += f"    static constexpr unsigned long long X_GEN = 0x{c.x_gen:016x}ULL;  /* xGen */\n"
        gw = words(w1 if gname == "g1" else w1 * w1)  # phi(x, y) = (GLV_W x, y) = [lambda](x, y): w on G1, w^2 on G2 (mulGLV, g1.go:536)
        out += f"    static constexpr uint32_t GLV_W[{len(gw)}] = {{" + ", ".join(f"0x{x:08x}u" for x in gw) + "};  /* Montgomery */\n"
        if kind in (1, 4):
            w = words(w1 if gname == "g1" else w1 * w1)
            out += f"    static constexpr uint32_t ENDO_W[{len(w)}] = {{" + ", ".join(f"0x{x:08x}u" for x in w) + "};  /* thirdRootOne of this group, Montgomery */\n"
        if kind in (2, 3):
            for nm, v in (("ENDO_U", c.endo_u), ("ENDO_V", c.endo_v)):
                ww = words(v[0]) + words(v[1])
                out += f"    static constexpr uint32_t {nm}[{len(ww)}] = {{" + ", ".join(f"0x{x:08x}u" for x in ww) + "};  /* psi: a0 then a1, Montgomery */\n"
        out += "};\n"
    return out


def render_cxx():
    s = "/* GENERATED by tools/gen_params.py -- do not edit. 32-bit-limb constexpr parameters for the HIP kernels.\n"
    s += " * Same numbers as gmsm_params.h (reference: ecc/<curve>/fp/element.go:44-74), split into 32-bit words. */\n"
    s += "#pragma once\n#include <stdint.h>\n\nnamespace gmsm {\n\n"
    for c in curves.CURVES.values():
        s += cxx_field(f"{c.name}_fp", c.p, c.fp_limbs)
        glv = curves.GlvParams(c)
        glv.check()
        s += cxx_field(f"{c.name}_fr", c.r, c.fr_limbs, (c.fr_root_of_unity, c.fr_max_order, c.fr_mult_gen), glv)
        s += group_consts(c)
        s += "\n"
    s += "}  // namespace gmsm\n"
    return s


def render(guard, who):
    s = f"/* GENERATED by tools/gen_params.py from gnark-crypto_amd/curves.py -- do not edit.\n"
    s += f" * Consumer: {who}. Constants are defined as in the reference's generated field packages\n"
    s += " * (ecc/bn254/fp/element.go:44-74 and the 6/12-limb twins). */\n"
    s += f"#ifndef {guard}\n#define {guard}\n#include <stdint.h>\n\n"
    for c in curves.CURVES.values():
        s += curve_block(c)
    s += f"#endif /* {guard} */\n"
    return s


def main():
    targets = [
        (os.path.join(ROOT, "gnark-crypto_amd", "csrc", "gmsm_params.h"), "GMSM_PARAMS_H", "HIP kernels / C-ABI host code"),
        (os.path.join(ROOT, "oracle", "oracle_params.h"), "ORACLE_PARAMS_H", "CPU oracle (test infrastructure)"),
    ]
    for path, guard, who in targets:
        txt = render(guard, who)
        with open(path, "w") as f:
            f.write(txt)
        print("wrote", path)
    path = os.path.join(ROOT, "gnark-crypto_amd", "csrc", "gmsm_params32.h")
    with open(path, "w") as f:
        f.write(render_cxx())
    print("wrote", path)


if __name__ == "__main__":
    main()
