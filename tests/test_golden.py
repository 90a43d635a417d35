"""Golden `.ctx` fixtures (tests/golden/, made by make_golden.py from the oracle): the oracle must
keep reproducing them (CPU), and so must the HIP path through the host program (GPU)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CASES = {"tiny_k31": (31, 2, ["alice", "bob"]), "tiny_k63": (63, 1, ["s63"]), "tiny_k5": (5, 1, ["undefined"])}


def _reads(name, c):
    return [l for l in open(os.path.join(GOLD, "%s.colour%d.txt" % (name, c))).read().split("\n")[:-1]]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_ctx(orc, name):
    k, ncols, names = CASES[name]
    g = orc.Graph(k, ncols, 1 << 14, seed=777)
    for c in range(ncols):
        if names[c] != "undefined":
            g.set_sample(c, names[c])
        b, o = orc.pack_reads(_reads(name, c))
        st = g.add_reads(c, b, o, nthreads=3)
        g.update_stats(c, st)
    assert g.ctx_bytes(True) == open(os.path.join(GOLD, name + ".ctx"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_k31", "tiny_k63"])
def test_cli_reproduces_golden_ctx(mcx, name, tmp_path):
    k, ncols, names = CASES[name]
    exe = os.path.join(os.path.dirname(HERE), "mccortex_amd", "bin", "mccortex%d" % (31 if k <= 31 else 63))
    args = [exe, "build", "-q", "-k", str(k), "-n", "64K", "--sort"]
    for c in range(ncols):
        args += ["--sample", names[c], "--seq", os.path.join(GOLD, "%s.colour%d.txt" % (name, c))]
    out = str(tmp_path / "o.ctx")
    p = subprocess.run(args + [out], stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert open(out, "rb").read() == open(os.path.join(GOLD, name + ".ctx"), "rb").read()
