#!/usr/bin/env python3
"""Regenerates the golden `.ctx` fixtures of tests/golden/ from the oracle (run from the repo root).

The reference tree holds no golden `.ctx` and its binary cannot be built in this
image, so these files freeze the ORACLE's output for fixed inputs: they guard the oracle (and,
through the parity tests, the GPU path) against regressions between rounds.  Inputs are stored
next to the outputs as plain text, one read per line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"tiny_k31": (31, 2, ["alice", "bob"]), "tiny_k63": (63, 1, ["s63"]), "tiny_k5": (5, 1, ["undefined"])}


def reads_for(name):
    g = synth.genome(3000, 5)
    out = []
    for c in range(2):
        b, o = synth.reads(120, 90, seed=40 + c, g=g, n_frac=0.1, lower_frac=0.1, var_len=True)
        out.append([bytes(b[int(o[i]):int(o[i + 1])]).decode() for i in range(len(o) - 1)])
    return out


if __name__ == "__main__":
    for name, (k, ncols, names) in CASES.items():
        rs = reads_for(name)
        g = orc.Graph(k, ncols, 1 << 14)
        for c in range(ncols):
            if names[c] != "undefined":
                g.set_sample(c, names[c])
            with open(os.path.join(HERE, "%s.colour%d.txt" % (name, c)), "w") as f:
                f.write("\n".join(rs[c]) + "\n")
            b, o = orc.pack_reads(rs[c])
            st = g.add_reads(c, b, o)
            g.update_stats(c, st)
        open(os.path.join(HERE, name + ".ctx"), "wb").write(g.ctx_bytes(True))
        print(name, g.nkmers, "k-mers")
