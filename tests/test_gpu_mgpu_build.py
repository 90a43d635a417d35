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


@pytest.mark.parametrize("use_v3", [True, False])
def test_sharded_inserter_empty_and_ragged_steps(mcx, orc, use_v3):
    """the step runner itself (1-rank RCCL group): steps of different sizes, empty steps (what a rank with a
    shorter share of a file submits), several insert() calls, two colours"""
    import torch
    import torch.distributed as dist
    from mccortex_amd import shard
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = "29811" if use_v3 else "29812"
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    try:
        k = 31
        dev = torch.device("cuda", 0)
        g = mcx.Graph(k, 2, 1 << 20) if use_v3 else mcx.Graph(k, 2, 1 << 20, nparts=1, part=0)
        og = orc.Graph(k, 2, 1 << 20)
        ins = shard.ShardedInserter(g, 1, dev, 400_000, use_v3)
        empty = torch.zeros(16, dtype=torch.uint8, device=dev)
        gen = synth.genome(20000, 3)
        plan = [(0, [1500, 0, 700, 0, 0, 2000]), (1, [0]), (1, [300, 2500]), (0, [50])]
        seed = 0
        for col, sizes in plan:
            steps = []
            for n in sizes:
                if n == 0:
                    steps.append((empty, 0))
                    continue
                seed += 1
                b, o = synth.reads(n, 120, seed=seed, g=gen, n_frac=0.05, var_len=True)
                og.add_reads(col, b, o)
                s = torch.from_numpy(synth.to_stream(b, o)).to(dev)
                steps.append((s, s.numel()))
            ins.insert(col, steps)
        assert g.nkmers == og.nkmers
        assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
        with pytest.raises(ValueError):
            big = torch.zeros(500_000, dtype=torch.uint8, device=dev)
            ins.insert(0, [(big, big.numel())])
        g.close()
    finally:
        if own_group:
            dist.destroy_process_group()
