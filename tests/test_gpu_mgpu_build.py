"""The multi-GPU build tool end to end on the one GPU of the test box (world size 1, RCCL backend):
reads -> per-owner bins -> all-to-all -> owners insert -> shards -> one .ctx, byte-identical to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from test_seqstream import _write

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, port, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "mccortex_amd.mgpu_build"] + args
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return p.returncode, p.stderr.decode(errors="replace")


@pytest.mark.parametrize("k,exchange", [(31, "v3"), (31, "v2"), (21, "v2"), (63, "v2")])
def test_tool_matches_oracle(mcx, orc, tmp_path, k, exchange):
    g = synth.genome(30000, 7)
    sets = []
    for i, (fmt, width) in enumerate([("fasta", 23), ("fastq", 0), ("plain", 0)]):
        bases, offs = synth.reads(2500, 110, seed=10 + i, g=g, n_frac=0.05, lower_frac=0.1, var_len=(fmt != "plain"))
        reads = [bytes(bases[int(offs[j]):int(offs[j + 1])]) for j in range(len(offs) - 1)]
        if fmt == "plain":
            reads = [r for r in reads if r]
        d = tmp_path / ("d%d" % i)
        d.mkdir()
        sets.append((_write(d, reads, fmt, width), reads))
    out = str(tmp_path / "out.ctx")
    args = ["-k", str(k), "-n", "1M", "--sort", "--step-bytes", "100K", "--sample", "alice", "--seq", sets[0][0], "--seq", sets[1][0],
            "--sample", "bob", "--seq2", sets[2][0] + ":" + sets[0][0], out]
    rc, err = _run(args, 29600 + k, env={"MCX_EXCHANGE": exchange})
    assert rc == 0, err[-3000:]
    og = orc.Graph(k, 2, 1 << 20)
    og.set_sample(0, "alice"); og.set_sample(1, "bob")
    for col, (path, reads) in zip((0, 0, 1, 1), sets + [sets[0]]):
        b, o = orc.pack_reads(reads)
        st = og.add_reads(col, b, o)
        og.update_stats(col, st)
    got = open(out, "rb").read()
    want = og.ctx_bytes(True)
    assert len(got) == len(want) and got == want
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    # refuses to overwrite without -f
    rc, err = _run(args, 29700 + k)
    assert rc != 0 and "already exists" in err
