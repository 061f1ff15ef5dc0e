"""Guard against an LLVM AMDGPU code-generation trap met in round 2: in a device FUNCTION (not a kernel) larger than the
+-128 KB reach of s_cbranch, the branch-relaxation pass expands a long branch as s_getpc_b64 / s_add_u32 / s_addc_u32 /
s_setpc_b64 on s[30:31] - the register pair that holds the function's RETURN ADDRESS in the calling convention. When such
a branch is taken (e.g. the early exit of a group addition whose operand is infinity in every lane of the wave) the
function later "returns" to the branch target and never comes back: the UnsatOpsMid policy hung exactly so for the 28-word
element types (one 330 KB addition body) and ran fine for BN254 G2 (below 128 KB). Kernels are not affected (no return).
usage: python tools/check_long_branch.py gnark-crypto_amd/csrc/build/group*.o   -> exit code 1 if any function is hit."""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_mix import disassemble  # noqa: E402


def main():
    bad = 0
    for obj in sys.argv[1:]:
        text = disassemble(obj)
        name, is_kernel_like, hits, size = None, False, 0, 0
        kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M))  # absent in objdump output: fall back below

        def flush():
            nonlocal bad
            if name and hits and not is_kernel_like:
                print(f"{obj}: {name[:110]}: {hits} long branch(es) through s[30:31] in a {size * 4 // 1024}+ KB function")
                bad += 1
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                flush()
                name, hits, size = m.group(1), 0, 0
                # kernels of this code base are the k_* templates (and hipcc's own __hip_* helpers have no returns)
                is_kernel_like = bool(re.search(r"(^|[0-9])k_[a-z]", name)) or name in kernels
                continue
            if name:
                size += 1
                if "s_getpc_b64 s[30:31]" in line:
                    hits += 1
        flush()
    print("long-branch check:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
