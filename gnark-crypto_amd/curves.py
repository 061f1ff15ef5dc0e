"""Curve parameters for the MultiExp path (BN254, BLS12-381, BW6-761).

Single source of truth for the host side; `tools/gen_params.py` turns it into the C headers used by
the HIP kernels and (separately) by the CPU oracle.  Values are the public curve constants, cited from
the reference:

* BN254     fp/fr moduli  ecc/bn254/fp/element.go:31, ecc/bn254/fr/element.go:31; generators ecc/bn254/bn254.go:103-123
* BLS12-381 fp/fr moduli  ecc/bls12-381/fp/element.go:31, fr/element.go:31; generators ecc/bls12-381/bls12-381.go:98-116
* BW6-761   fp/fr moduli  ecc/bw6-761/fp/element.go:31, fr/element.go:31; generators ecc/bw6-761/bw6-761.go:91-104

All three curves are y^2 = x^3 + b with a = 0; G2 of BN254/BLS12-381 lives over Fp2 = Fp[u]/(u^2+1),
G2 of BW6-761 lives over Fp.  `tests/test_params.py` re-derives every Montgomery constant and, when the
reference tree is present, compares them with the reference's generated constants.
"""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class Curve:
    name: str            # C identifier: bn254, bls12_381, bw6_761
    p: int               # base-field modulus
    r: int               # scalar-field modulus (group order)
    g1: Tuple[int, int]  # affine generator of G1
    g2: tuple            # affine generator of G2 ((x0,x1),(y0,y1)) over Fp2, or (x,y) over Fp
    g2_ext: int          # extension degree of the G2 coordinate field (2 or 1)
    b: int               # curve coefficient (only used by tests to check on-curve-ness)
    fr_root_of_unity: int = 0   # primitive 2^fr_max_order-th root of unity of the scalar field (fr/generator.go:23)
    fr_max_order: int = 0       # fr/generator.go:24
    fr_mult_gen: int = 0        # generator of Fr^* used as the coset shift (fr/fft/domain.go:56-62)
    # endomorphism constants of the reference's IsInSubGroup tests (ecc/<curve>/<curve>.go init()):
    x_gen: int = 0              # |x|, the curve's seed as the reference stores it (xGen)
    third_root_one_g1: int = 0  # w with w^3 = 1: phi(x, y) = (w x, y) on G1; G2 uses w^2 (thirdRootOneG2)
    endo_u: tuple = ()          # psi(x, y) = (conj(x) u, conj(y) v) on G2 over Fp2 (endo.u, endo.v as (a0, a1))
    endo_v: tuple = ()
    lambda_glv: int = 0         # phi(P) = [lambda]P on the r-torsion (lambdaGLV): lambda^2 + lambda + 1 = 0 mod r

    @property
    def fp_limbs(self) -> int:      # 64-bit limbs of an fp.Element
        return (self.p.bit_length() + 63) // 64

    @property
    def fr_limbs(self) -> int:      # 64-bit limbs of an fr.Element
        return (self.r.bit_length() + 63) // 64

    @property
    def fr_bits(self) -> int:
        return self.r.bit_length()

    @property
    def fp_R(self) -> int:          # Montgomery radix of fp
        return 1 << (64 * self.fp_limbs)

    @property
    def fr_R(self) -> int:
        return 1 << (64 * self.fr_limbs)


BN254 = Curve(
    name="bn254",
    p=0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    r=0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    g1=(1, 2),
    g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    g2_ext=2,
    b=3,
    fr_root_of_unity=19103219067921713944291392827692070036145651957329286315305642004821462161904,
    fr_max_order=28,
    fr_mult_gen=5,
    x_gen=4965661367192848881,  # bn254.go:146
    lambda_glv=4407920970296243842393367215006156084916469457145843978461,  # bn254.go:133
    third_root_one_g1=2203960485148121921418603742825762020974279258880205651966,  # bn254.go:131
    endo_u=(21575463638280843010398324269430826099269044274347216827212613867836435027261,  # bn254.go:137-140
            10307601595873709700152284273816112264069230130616436755625194854815875713954),
    endo_v=(2821565182194536844548159561693502659359617185244120367078079554186484126554,
            3505843767911556378687030309984248845540243509899259641013678093033130930403),
)

BLS12_381 = Curve(
    name="bls12_381",
    p=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    g1=(3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
        1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569),
    g2=((352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
         3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
        (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
         927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582)),
    g2_ext=2,
    b=4,
    fr_root_of_unity=10238227357739495823651030575849232062558860180284477541189508159991286009131,
    fr_max_order=32,
    fr_mult_gen=7,
    lambda_glv=228988810152649578064853576960394133503,  # bls12-381.go:128
    x_gen=15132376222941642752,  # bls12-381.go:141 (the seed is -x_gen; the tests below are written for the stored value)
    third_root_one_g1=4002409555221667392624310435006688643935503118305586438271171395842971157480381377015405980053539358417135540939436,  # :126
    endo_u=(0, 4002409555221667392624310435006688643935503118305586438271171395842971157480381377015405980053539358417135540939437),  # :132-135
    endo_v=(2973677408986561043442465346520108879172042883009249989176415018091420807192182638567116318576472649347015917690530,
            1028732146235106349975324479215795277384839936929757896155643118032610843298655225875571310552543014690878354869257),
)

BW6_761 = Curve(
    name="bw6_761",
    p=0x122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b,
    r=0x1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001,
    g1=(6238772257594679368032145693622812838779005809760824733138787810501188623461307351759238099287535516224314149266511977132140828635950940021790489507611754366317801811090811367945064510304504157188661901055903167026722666149426237,
        2101735126520897423911504562215834951148127555913367997162789335052900271653517958562461315794228241561913734371411178226936527683203879553093934185950470971848972085321797958124416462268292467002957525517188485984766314758624099),
    g2=(6445332910596979336035888152774071626898886139774101364933948236926875073754470830732273879639675437155036544153105017729592600560631678554299562762294743927912429096636156401171909259073181112518725201388196280039960074422214428,
        562923658089539719386922163444547387757586534741080263946953401595155211934630598999300396317104182598044793758153214972605680357108252243146746187917218885078195819486220416605630144001533548163105316661692978285266378674355041),
    g2_ext=1,
    b=-1,
    fr_root_of_unity=32863578547254505029601261939868325669770508939375122462904745766352256812585773382134936404344547323199885654433,
    fr_max_order=46,
    fr_mult_gen=15,
    x_gen=9586122913090633729,  # bw6-761.go:128
    lambda_glv=80949648264912719408558363140637477264845294720710499478137287262712535938301461879813459410945,  # bw6-761.go:123
    third_root_one_g1=1968985824090209297278610739700577151397666382303825728450741611566800370218827257750865013421937292370006175842381275743914023380727582819905021229583192207421122272650305267822868639090213645505120388400344940985710520836292650,  # :121
)

CURVES = {c.name: c for c in (BN254, BLS12_381, BW6_761)}


# ---- GLV scalar decomposition (ecc/utils.go:62-170: PrecomputeLattice / SplitScalar), restated for the device ----
def glv_lattice(r, lam):
    """Two short vectors (a, b) with a + b lam = 0 mod r, by the extended Euclidean walk of PrecomputeLattice
    (ecc/utils.go:62-122): the remainder sequence of (r, lam) is followed until it drops below sqrt(r)."""
    from math import isqrt
    rst = [[r, 1, 0], [lam, 0, 1]]
    root = isqrt(r)
    while rst[1][0] >= root:
        q, rem = divmod(rst[0][0], rst[1][0])
        rst[0], rst[1] = rst[1], [rem, rst[0][1] - rst[1][1] * q, rst[0][2] - rst[1][2] * q]
    q, rem = divmod(rst[0][0], rst[1][0])
    t = rst[0][2] - rst[1][2] * q
    v1 = (rst[1][0], -rst[1][2])
    v2 = (rem, -t) if rst[0][0] ** 2 + rst[0][2] ** 2 > rem * rem + t * t else (rst[0][0], -rst[0][2])
    return v1, v2


class GlvParams:
    """What the device needs to split s into (k1, k2) with k1 + k2 lam = s mod r and |k1|, |k2| < 2^bits:
        m1 = (s |b1|) >> 32 sh,  m2 = (s |b2|) >> 32 sh          (b_i = round(2^(32 sh) v_2i / det), SplitScalar's b1 / b2)
        k1 = s - m1 a11 - m2 a21,  k2 = - m1 a12 - m2 a22        mod 2^(32 hl), read as two's complement
    with a11 = sgn(b1) v11, a12 = sgn(b1) v12, a21 = -sgn(b2) v21, a22 = -sgn(b2) v22.  ANY integers m1, m2 give the
    congruence (v1, v2 are lattice vectors); the roundings only decide how short k1, k2 are: (k1, k2) = e1 v1 + e2 v2 with
    |e_i| < 1 + 2^-32, hence |k1| <= |v11| + |v21|, |k2| <= |v12| + |v22| (checked below on the extreme and on random s)."""

    def __init__(self, c):
        r, lam = c.r, c.lambda_glv
        assert (lam * lam + lam + 1) % r == 0
        v1, v2 = glv_lattice(r, lam)
        assert (v1[0] + v1[1] * lam) % r == 0 and (v2[0] + v2[1] * lam) % r == 0
        det = v1[0] * v2[1] - v1[1] * v2[0]
        assert abs(det) == r
        nr = 2 * c.fr_limbs                      # 32-bit limbs of a scalar
        self.sh = nr + 1                         # 32 more bits than the scalar: the rounding of b costs < 2^-32
        rnd = lambda a, d: (2 * a + d) // (2 * d) if d > 0 else (2 * -a + -d) // (2 * -d)   # nearest integer to a / d
        b1 = rnd(v2[1] << (32 * self.sh), det)
        b2 = rnd(v1[1] << (32 * self.sh), det)
        sg = lambda x: -1 if x < 0 else 1
        self.b1, self.b2 = abs(b1), abs(b2)
        self.a = [sg(b1) * v1[0], -sg(b2) * v2[0], sg(b1) * v1[1], -sg(b2) * v2[1]]  # a11, a21, a12, a22
        bound = max(abs(v1[0]) + abs(v2[0]), abs(v1[1]) + abs(v2[1])) + 1
        self.bits = bound.bit_length()
        self.hl = (self.bits + 1 + 31) // 32     # + sign bit
        self.nb = (max(self.b1, self.b2).bit_length() + 31) // 32
        self.r, self.lam = r, lam
        assert (r * max(self.b1, self.b2)) >> (32 * self.sh) < 1 << (32 * self.hl)  # m1, m2 fit the half width

    def split(self, s):
        """(k1, k2) exactly as the device computes them (signed Python ints)."""
        mod = 1 << (32 * self.hl)
        m1 = (s * self.b1) >> (32 * self.sh)
        m2 = (s * self.b2) >> (32 * self.sh)
        k1 = (s - m1 * self.a[0] - m2 * self.a[1]) % mod
        k2 = (-m1 * self.a[2] - m2 * self.a[3]) % mod
        signed = lambda k: k - mod if k >> (32 * self.hl - 1) else k
        return signed(k1), signed(k2)

    def check(self, samples=2000):
        import random
        rng = random.Random(0x676C76)
        cases = [0, 1, 2, self.r - 1, self.r - 2, self.r // 2, self.r // 3, self.lam, self.lam + 1, self.r - self.lam, (1 << (self.r.bit_length() - 1))]
        cases += [rng.randrange(self.r) for _ in range(samples)]
        worst = 0
        for s in cases:
            k1, k2 = self.split(s)
            assert (k1 + k2 * self.lam - s) % self.r == 0
            worst = max(worst, abs(k1), abs(k2))
        assert worst < 1 << self.bits, (worst.bit_length(), self.bits)
        return worst.bit_length()
