#!/usr/bin/env python3
"""Generates tests/golden/golden_lvl2_v1.json from the CPU oracle (oracle/tfhe_oracle_lvl2.c).

Same role as make_golden.py for the N = 2048 / 64-bit-torus path: seeds -> keys (sha256 of
every key array), seeded lvl0 input ciphertexts, and for each of the 14 ops the sha256 of the
output words (full words for NAND and MUX, gate 0).  The reference has no N = 2048 path, so
these vectors pin the oracle against itself over time and the HIP path against the oracle.
    python tests/golden/make_golden_lvl2.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

KEY_SEED, KEY2_SEED = 1, 7
TRIPLES = [(0, 0, 1), (0, 1, 0), (1, 0, 1), (1, 1, 0)]
FULL = {"NAND", "MUX"}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def inputs(keys):
    bits = np.array(TRIPLES, np.uint8)
    return bits, [keys.encrypt(bits[:, i], 0, seed=7000 + i) for i in range(3)]


def main():
    L = ol.load()
    keys = ol.Keys(L, seed=KEY_SEED)
    keys2 = ol.KeysLvl2(L, keys, seed=KEY2_SEED)
    bits, ins = inputs(keys)
    g = {"version": 1, "key_seed": KEY_SEED, "key2_seed": KEY2_SEED, "triples": TRIPLES,
         "params": {"n": ol.n, "N": ol.N2, "l": 4, "Bgbit": 9, "t": 7, "basebit": 2, "mu": ol.MU2},
         "keys_sha256": {"s0": sha(keys.s0), "s2": sha(keys2.s2), "bk": sha(keys2.bk), "ksk": sha(keys2.ksk)},
         "inputs_sha256": [sha(x) for x in ins], "ops": {}}
    for op, name in enumerate(ol.OPS):
        out = keys2.gate_batch(op, ins[0], ins[1], ins[2])
        dec = keys.decrypt(out, 0)
        assert list(dec) == [ol.truth(L, op, *t) for t in TRIPLES], name
        entry = {"out_sha256": sha(out), "decrypt": [int(x) for x in dec]}
        if name in FULL:
            entry["out_words_gate0"] = [int(x) for x in out[0]]
        g["ops"][name] = entry
    # one accumulator after 3 CMux steps: pins the 64-bit external product itself
    acc = keys2.blind_rotate(ins[0][0], 3)
    g["acc_after_3_steps_sha256"] = sha(acc)
    with open(os.path.join(HERE, "golden_lvl2_v1.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("wrote golden_lvl2_v1.json")


if __name__ == "__main__":
    main()
