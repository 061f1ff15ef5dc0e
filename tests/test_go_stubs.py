"""What can be checked of the Go side of the boundary without a Go toolchain (integration/go/<curve>/): the file-level
build tags exclude each other, package names match the reference's packages, every C symbol a stub calls is declared in
include/gmsm.h and exported by libgmsm.so, the three curve directories are the same file up to the documented
substitutions, and the reference really defines the methods the recipe renames at the cited lines (when the reference
tree is present - it is not on the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "integration", "go")
CURVES = {"bn254": ("bn254", "gmsm_bn254_", "ecc/bn254", (32, 357)),
          "bls12-381": ("bls12381", "gmsm_bls12_381_", "ecc/bls12-381", (32, 355)),
          "bw6-761": ("bw6761", "gmsm_bw6_761_", "ecc/bw6-761", (32, 306))}


def read(curve, name):
    with open(os.path.join(GO, curve, name)) as f:
        return f.read()


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_stub_files_are_consistent(gm, curve):
    pkg, prefix, path, _ = CURVES[curve]
    dev, pure = read(curve, "multiexp_mi355x.go"), read(curve, "multiexp_purego.go")
    assert dev.startswith("//go:build mi355x\n") and pure.startswith("//go:build !mi355x\n")  # file-level, mutually exclusive
    for text in (dev, pure):
        assert re.search(rf"^package {pkg}$", text, re.M)
        assert f'"github.com/consensys/gnark-crypto/{path}/fr"' in text
        for recv in ("G1Jac", "G2Jac"):  # both files define exactly the exported method pair
            assert len(re.findall(rf"^func \(p \*{recv}\) MultiExp\(", text, re.M)) == 1
    assert dev.count("p.multiExpCPU(points, scalars, config)") == 2 and pure.count("p.multiExpCPU(points, scalars, config)") == 2
    header = open(os.path.join(ROOT, "include", "gmsm.h")).read()
    lib = gm._lib.load()
    called = set(re.findall(r"C\.(gmsm_[a-z0-9_]+)\(", dev))
    assert {prefix + "g1_multiexp", prefix + "g2_multiexp", "gmsm_last_error"} <= called
    for sym in called:
        assert re.search(rf"\b{sym}\s*\(", header), sym
        assert hasattr(lib, sym), sym
    # the same file as bn254's up to the documented substitutions
    ref_pkg, ref_prefix, ref_path, ref_lines = CURVES["bn254"]
    for name in ("multiexp_mi355x.go", "multiexp_purego.go"):
        base = read("bn254", name).replace(f"package {ref_pkg}", f"package {pkg}").replace(ref_prefix, prefix)
        base = base.replace(f"{ref_path}/fr", f"{path}/fr")
        mine = read(curve, name)
        strip = lambda t: re.sub(r"//.*", "", t)  # comments cite per-curve line numbers
        assert strip(base) == strip(mine), name


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_recipe_matches_the_reference(curve):
    pkg, _, path, lines = CURVES[curve]
    ref = os.path.join("/root/reference", path, "multiexp.go")
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    src = open(ref).read().splitlines()
    assert src[lines[0] - 1].startswith("func (p *G1Jac) MultiExp(")
    assert src[lines[1] - 1].startswith("func (p *G2Jac) MultiExp(")
    assert any(re.match(rf"^package {pkg}$", ln) for ln in src[:12])
    tmpl = open("/root/reference/internal/generator/ecc/template/multiexp.go.tmpl").read().splitlines()
    assert tmpl[247 - 1].startswith("func (p *{{ $.TJacobian }}) MultiExp(")
    assert "_p.MultiExp(points[:nbPoints/2]" in tmpl[350 - 1] and "p.MultiExp(points[nbPoints/2:]" in tmpl[353 - 1]


# ---- N1 on the Go side: the resident proving key of package kzg (integration/go/<curve>/kzg/)
KZG_SUBST = {"bn254": ("bn254", "GMSM_BN254_G1"), "bls12-381": ("bls12381", "GMSM_BLS12_381_G1"), "bw6-761": ("bw6761", "GMSM_BW6_761_G1")}


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_kzg_resident_key_files(gm, curve):
    pkg_alias, group_const = KZG_SUBST[curve]
    path = CURVES[curve][2]
    dev, pure = read(curve, os.path.join("kzg", "kzg_mi355x.go")), read(curve, os.path.join("kzg", "kzg_purego.go"))
    assert dev.startswith("//go:build mi355x\n") and pure.startswith("//go:build !mi355x\n")
    api = ["func NewResidentProvingKey(pk ProvingKey, windowTables bool) (*ResidentProvingKey, error)",
           "func ReadDumpResident(path string, windowTables bool, maxPkPoints ...int) (*ResidentProvingKey, *VerifyingKey, error)",
           "func ReadFromResident(r io.Reader, windowTables bool, subgroupCheck bool) (*ResidentProvingKey, int64, error)",
           "func (rk *ResidentProvingKey) Size() int", "func (rk *ResidentProvingKey) Commit(p []fr.Element, nbTasks ...int) (Digest, error)",
           "func (rk *ResidentProvingKey) CommitBatch(ps [][]fr.Element) ([]Digest, error)", "func (rk *ResidentProvingKey) Release()"]
    for text in (dev, pure):
        assert re.search(r"^package kzg$", text, re.M)
        assert f'"github.com/consensys/gnark-crypto/{path}/fr"' in text
        for sig in api:  # the same exported API in both builds
            assert text.count(sig) == 1, sig
    assert f'"github.com/consensys/gnark-crypto/{path}"' in dev and f"{pkg_alias}.G1Jac" in dev and f"C.{group_const}" in dev
    assert "runtime.SetFinalizer" in dev and "Commit(p, rk.host, nbTasks...)" in dev and "len(p) < MinDevicePoints" in dev
    header = open(os.path.join(ROOT, "include", "gmsm.h")).read()
    lib = gm._lib.load()
    called = set(re.findall(r"C\.(gmsm_[a-z0-9_]+)\(", dev))
    assert {"gmsm_bases_register", "gmsm_bases_register_dump", "gmsm_bases_register_compressed", "gmsm_bases_register_raw", "gmsm_bases_precompute", "gmsm_bases_release", "gmsm_multiexp_bases",
            "gmsm_multiexp_bases_batch", "gmsm_last_error"} <= called
    for sym in called:
        assert re.search(rf"\b{sym}\s*\(", header), sym
        assert hasattr(lib, sym), sym
    assert re.search(rf"\b{group_const}\b", header)
    # argument counts of the calls match the prototypes (a cgo call with the wrong arity does not compile)
    decls = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # prototypes only: the comments mention the functions too
    for sym in called:
        proto = re.search(rf"\b{sym}\s*\(([^;]*?)\)\s*;", decls, re.S).group(1)
        nargs = 0 if proto.strip() in ("", "void") else proto.count(",") + 1
        for m in re.finditer(rf"C\.{sym}\(", dev):
            depth, i, commas = 1, m.end(), 0
            while depth:
                ch = dev[i]
                depth += ch == "("
                depth -= ch == ")"
                commas += (ch == "," and depth == 1)
                i += 1
            inner = dev[m.end():i - 1].strip()
            assert (0 if not inner else commas + 1) == nargs, (sym, inner)
    # the same files as bn254's up to the documented substitutions
    for name in ("kzg_mi355x.go", "kzg_purego.go"):
        base = read("bn254", os.path.join("kzg", name)).replace("ecc/bn254", path).replace("bn254.", pkg_alias + ".").replace("GMSM_BN254_G1", group_const)
        strip = lambda t: re.sub(r"//.*", "", t)
        assert strip(base) == strip(read(curve, os.path.join("kzg", name))), name


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_kzg_files_cite_the_reference(curve):
    path = CURVES[curve][2]
    ref = os.path.join("/root/reference", path, "kzg")
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    kzg = open(os.path.join(ref, "kzg.go")).read().splitlines()
    assert kzg[159 - 1].startswith("func Commit(p []fr.Element, pk ProvingKey, nbTasks ...int) (Digest, error)")
    assert "res.MultiExp(pk.G1[:len(p)], p, config)" in "\n".join(kzg[159:176])
    assert kzg[180 - 1].startswith("func Open(") and kzg[246 - 1].startswith("func BatchOpenSinglePoint(")
    assert any("ErrInvalidPolynomialSize" in ln for ln in kzg[:40]) and any("ErrMinSRSSize" in ln for ln in kzg[:40])
    marshal = open(os.path.join(ref, "marshal.go")).read().splitlines()
    assert marshal[98 - 1].startswith("func (srs *SRS) ReadDump(") and "unsafe.ReadSlice" in "\n".join(marshal[98:113])
    # ReadFromResident's twins and the stream they read
    assert marshal[16 - 1].startswith("func (pk *ProvingKey) WriteTo(") and "enc.Encode(pk.G1)" in "\n".join(marshal[16:32])
    assert marshal[140 - 1].startswith("func (pk *ProvingKey) ReadFrom(") and "dec.Decode(&pk.G1)" in "\n".join(marshal[140:147])
    assert marshal[151 - 1].startswith("func (pk *ProvingKey) UnsafeReadFrom(") and "NoSubgroupChecks()" in "\n".join(marshal[151:158])
    dec = open(os.path.join("/root/reference", path, "marshal.go")).read()
    assert "unsafeComputeY(dec.subGroupCheck)" in dec and "func isCompressed(msb byte) bool" in dec
    g1 = open(os.path.join("/root/reference", path, "g1.go")).read()
    assert "func BatchJacobianToAffineG1(points []G1Jac) []G1Affine" in g1
