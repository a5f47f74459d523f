#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.json from the CPU oracle (oracle/tfhe_oracle.c).

The reference holds no golden vectors for this path and cannot run here, so
these vectors pin the oracle against ITSELF across time and pin the HIP path against the
oracle on the GPU box without recomputing it: seeds -> keys (sha256 of every key array),
seeded input ciphertexts, and for each of the 14 ops x 2 levels the sha256 of the output
words plus the full words of a few outputs.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

KEY_SEED = 1
TRIPLES = [(0, 0, 1), (0, 1, 0), (1, 0, 1), (1, 1, 0)]     # (in0, in1, in2) plaintext bits
FULL = {"NAND", "MUX"}                                      # ops whose output words are stored in full


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()


def inputs(keys, level):
    bits = np.array(TRIPLES, np.uint8)
    return bits, [keys.encrypt(bits[:, i], level, seed=5000 + 100 * level + i) for i in range(3)]


def main():
    L = ol.load()
    keys = ol.Keys(L, seed=KEY_SEED)
    g = {"version": 1, "key_seed": KEY_SEED, "triples": TRIPLES,
         "params": {"n": ol.n, "N": ol.N, "l": 3, "Bgbit": 6, "t": 8, "basebit": 2, "mu": ol.MU},
         "keys_sha256": {"s0": sha(keys.s0), "s1": sha(keys.s1), "bk": sha(keys.bk), "ksk": sha(keys.ksk)},
         "levels": {}}
    for level in (0, 1):
        bits, ins = inputs(keys, level)
        lv = {"inputs_sha256": [sha(x) for x in ins], "ops": {}}
        for op, name in enumerate(ol.OPS):
            out = keys.gate_batch(op, level, ins[0], ins[1], ins[2])
            dec = keys.decrypt(out, level)
            exp = [ol.truth(L, op, *t) for t in TRIPLES]
            assert list(dec) == exp, (name, level)
            entry = {"out_sha256": sha(out), "decrypt": [int(x) for x in dec]}
            if name in FULL:
                entry["out_words_gate0"] = [int(x) for x in out[0]]
            lv["ops"][name] = entry
        g["levels"][str(level)] = lv
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("wrote golden_v1.json")


if __name__ == "__main__":
    main()
