"""The multi-GPU build tool end to end on the one GPU of the test box:
reads -> per-owner bins -> all-to-all -> owners insert -> shards -> one .ctx, byte-identical to the oracle.
World size 1 runs under the RCCL backend; world size 2 / 4 -- rank 1 exists: read slicing, step-count agreement,
per-rank statistics, shard merge, header-once -- runs as N processes sharing cuda:0 through shard.py's test
transport (MCX_DIST_BACKEND=gloo: RCCL refuses two ranks on one device), with the real kernels on every rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from test_seqstream import _write

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ONE_GPU = {"MCX_DIST_BACKEND": "gloo", "MCX_DIST_ONE_DEVICE": "0"}


def _launch(world, port, tail, env=None, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + tail
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if world > 1:
        e.update(ONE_GPU)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if p.returncode != 0 and b"already exists" not in p.stderr:   # the whole log of a failed launch, for the builder (gpurun_out/ is scratch)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "failed_launch_%d.log" % port), "wb") as f:
            f.write(p.stdout + b"\n---- stderr ----\n" + p.stderr)
    return p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")


def _run(args, port, env=None, world=1):
    rc, _, err = _launch(world, port, ["-m", "mccortex_amd.mgpu_build"] + args, env)
    return rc, err


@pytest.mark.parametrize("k,exchange,world", [(31, "v3", 1), (31, "v2", 1), (21, "v2", 1), (63, "v2", 1),
                                              (31, "v3", 2), (31, "v2", 2), (63, "v3", 2), (63, "v2", 2), (21, "v2", 2), (31, "v3", 4)])
def test_tool_matches_oracle(mcx, orc, tmp_path, k, exchange, world):
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
    rc, err = _run(args, 29600 + k + 100 * world, env={"MCX_EXCHANGE": exchange}, world=world)
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
    rc, err = _run(args, 29700 + k, world=world)
    assert rc != 0 and "already exists" in err


def test_bench_two_ranks_strong_reproduces_the_single_gpu_graph(mcx, tmp_path):
    """`python bench.py --gpus 2` AS THE DRIVER TYPES IT -- a plain subprocess, no launcher: bench.py starts its two ranks
    itself -- on cuda:0 through the test transport.  Strong scaling = BASELINE config C3 (the N = 1 reads dealt out to the
    ranks, the N = 1 table split 2 ways): one JSON line that times BOTH exchange formats and the C5 pass (4 colours);
    the sum of the ranks' graph checksums and node counts must be what the N = 1 run of the same steps reports in either
    format, and the coloured graph must be the one a single rank builds."""
    import json
    # (--defer-tuples: two ranks and this pytest process share the one device's HBM here)
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--batch-reads", "1000000", "--defer-tuples", "1000000000"]
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)

    def bench(n, env=None):
        p = subprocess.run([sys.executable, "bench.py", "--gpus", str(n)] + common, cwd=ROOT, env=dict(e, **(env or {})),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
        lines = p.stdout.decode().strip().splitlines()
        assert len(lines) == 1, lines[:3]   # ONE JSON line on stdout
        return json.loads(lines[0])

    one = bench(1)
    ref = bench(1, {"MCX_BENCH_FORCE_SHARD": "1", "MASTER_PORT": "29853"})   # the N > 1 code with one rank (RCCL): its C5 graph is the reference
    two = bench(2, ONE_GPU)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["steps"] == 2
    cfg = two["config"]
    assert set(cfg["exchange_formats"]) >= {"v3", "v2"} and cfg["formats_agree"] is True
    for fmt in ("v3", "v2"):
        r = cfg["exchange_formats"][fmt]
        assert r["graph_checksum"] == one["config"]["graph_checksum"], fmt
        assert r["distinct_kmers_total"] == one["config"]["distinct_kmers_total"]
        assert r["kmers_inserted"] == one["config"]["kmers_inserted"]
        assert r["fallback_inserts_total"] == 0
        ranks = r["per_rank"]
        assert [x["rank"] for x in ranks] == [0, 1]
        assert sum(x["kmers_kmerised"] for x in ranks) == one["config"]["kmers_inserted"]
        for x in ranks:
            assert x["link_bytes_sent"] > 0 and x["exchange_steps"] == 1 and {"sender", "insert"} <= set(x["stage_ms"])
    assert cfg["exchange_formats"]["v3"]["link_bytes_per_occurrence"] < cfg["exchange_formats"]["v2"]["link_bytes_per_occurrence"]
    assert two["value"] == max(cfg["exchange_formats"][f]["value"] for f in ("v3", "v2"))
    assert two["multi_gpu"]["exchange_format"] == cfg["exchange_format"] and cfg["summary"]["formats_agree"] is True
    c5 = cfg["C5"]
    assert c5["colours"] == 4 and c5["nodes_and_kmers_match_one_colour"] is True
    assert [x["rank"] for x in c5["per_rank"]] == [0, 1]
    rc5 = ref["config"]["C5"]
    assert ref["config"]["graph_checksum"] == one["config"]["graph_checksum"]
    assert (c5["graph_checksum"], c5["distinct_kmers_total"], c5["kmers_inserted"]) == (rc5["graph_checksum"], rc5["distinct_kmers_total"], rc5["kmers_inserted"])
    assert c5["graph_checksum"] != one["config"]["graph_checksum"]   # (colours are part of a record)


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
