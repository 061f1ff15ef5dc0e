"""Instruction mix of one kernel of a built object, without a GPU.
usage: python tools/isa_mix.py gnark-crypto_amd/csrc/build/group0.o k_accumulate_segINS_3FpUINS_15bn254 [lo-hi,lo-hi,...]
Prints the basic blocks of the kernel (address, instructions, multiplier ops, memory ops, closing branch) and the mix
of the whole kernel; with address ranges (hex, as printed in the block table) also the mix of just those blocks - the
common path of a loop iteration is picked by hand from the table (profiles/r02_accumulate_isa.md)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def kernel_instructions(text, name):
    ins, on = [], False
    for line in text.splitlines():
        if re.match(r"^[0-9a-f]+ <", line):
            on = name in line
            continue
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if on and m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def mix(ins, title):
    c = collections.Counter(op for _, op, _ in ins)
    mul = c["v_mad_u64_u32"] + c["v_mul_lo_u32"]
    print(f"{title}: {len(ins)} instructions, {mul} multiplier ops ({100.0 * mul / max(1, len(ins)):.1f} %)")
    for op, n in c.most_common(25):
        print(f"  {op:28s} {n:6d}  {100.0 * n / len(ins):5.1f} %")


def main():
    obj, name = sys.argv[1], sys.argv[2]
    ins = kernel_instructions(disassemble(obj), name)
    if not ins:
        raise SystemExit("kernel not found")
    is_branch = lambda op: op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm")
    targets = set()
    for a, op, args in ins:
        m = re.match(r"\s*(\d+)", args)
        if is_branch(op) and op != "s_endpgm" and m:
            simm = int(m.group(1))
            targets.add(a + 4 + (simm - 65536 if simm >= 32768 else simm) * 4)
    addrs = {a for a, _, _ in ins}
    bounds = sorted({ins[0][0]} | (targets & addrs) | {ins[i + 1][0] for i, x in enumerate(ins[:-1]) if is_branch(x[1])})
    for i, b in enumerate(bounds):
        e = bounds[i + 1] if i + 1 < len(bounds) else ins[-1][0] + 4
        blk = [x for x in ins if b <= x[0] < e]
        c = collections.Counter(op for _, op, _ in blk)
        print(f"{b:#x}  n={len(blk):5d}  mul={c['v_mad_u64_u32'] + c['v_mul_lo_u32']:5d}  "
              f"mem={sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'ds_'))):3d}  {blk[-1][1]}")
    mix(ins, "whole kernel")
    if len(sys.argv) > 3:
        sel = []
        for r in sys.argv[3].split(","):
            lo, hi = (int(v, 16) for v in r.split("-"))
            sel += [x for x in ins if lo <= x[0] < hi]
        mix(sel, "selected blocks")


if __name__ == "__main__":
    main()
