#!/usr/bin/env python3
"""Extract the RFC 9380 hash-to-curve known-answer points from the reference's own test files into a small JSON
fixture (the reference tree does not exist on the GPU box, so the vectors travel as a fixture).

Source:  /root/reference/ecc/bn254/hash_vectors_test.go, /root/reference/ecc/bls12-381/hash_vectors_test.go
         (hashToG1Vector / hashToG2Vector: msg, P, Q0, Q1 with P = clear_cofactor(Q0 + Q1))
Output:  tests/golden/hash_vectors.json
Run:     python tests/golden/make_hash_vectors.py   (only where /root/reference exists)
"""
import json
import os
import re

REF = "/root/reference/ecc"
HERE = os.path.dirname(os.path.abspath(__file__))

PT = r'point\{\s*"([^"]+)",\s*"([^"]+)",?\s*\}'
MSG = re.compile(r'msg:\s*"([^"]*)"')


def field(block, name):
    m = re.search(r'\b' + name + r':\s*' + PT, block)
    return m.group(1), m.group(2)


def coord(s):
    parts = [int(x, 16) for x in s.split(",")]
    return [hex(p) for p in parts]  # 1 entry for Fp, 2 (A0, A1) for Fp2


def extract(path):
    txt = open(path).read()
    out = {}
    for name in ("hashToG1Vector", "hashToG2Vector"):
        start = txt.index(name + " = hashTestVector")
        nxt = [txt.find(k, start + 10) for k in ("encodeToG1Vector =", "encodeToG2Vector =", "hashToG1Vector =", "hashToG2Vector =")]
        nxt = [k for k in nxt if k > start]
        block = txt[start:min(nxt) if nxt else len(txt)]
        cases = []
        msgs = list(MSG.finditer(block))
        for k, m in enumerate(msgs):  # one case = text from this msg to the next (field order differs between curves)
            sub = block[m.start():msgs[k + 1].start() if k + 1 < len(msgs) else len(block)]
            msg = m.group(1)
            (px, py), (q0x, q0y), (q1x, q1y) = field(sub, "P"), field(sub, "Q0"), field(sub, "Q1")
            cases.append({"msg": msg, "P": [coord(px), coord(py)], "Q0": [coord(q0x), coord(q0y)], "Q1": [coord(q1x), coord(q1y)]})
        out["g1" if "G1" in name else "g2"] = cases
    return out


def main():
    data = {
        "_source": "ecc/bn254/hash_vectors_test.go and ecc/bls12-381/hash_vectors_test.go of gnark-crypto @ 2025-01-17",
        "bn254": extract(os.path.join(REF, "bn254", "hash_vectors_test.go")),
        "bls12_381": extract(os.path.join(REF, "bls12-381", "hash_vectors_test.go")),
    }
    with open(os.path.join(HERE, "hash_vectors.json"), "w") as f:
        json.dump(data, f, indent=1)
    for c in ("bn254", "bls12_381"):
        print(c, {k: len(v) for k, v in data[c].items()})


if __name__ == "__main__":
    main()
