"""k = 65 .. 127: three- and four-word BinaryKmers (the reference's MAXK = 95 / 127 builds, tests/run.sh:31-34,
Makefile:33-48).  The fused kernel, record load, export, checksum, sort and the CLI against the oracle, byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import synth
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mccortex_amd", "bin")
WIDE_KS = [65, 67, 79, 93, 95, 97, 99, 111, 125, 127]


@pytest.mark.parametrize("k", WIDE_KS)
def test_wide_random_reads_match_oracle(mcx, orc, k):
    bases, offs = synth.reads(1500, 300, genome_len=30000, seed=k, n_frac=0.05, lower_frac=0.1)
    _compare(mcx, orc, k, 1, [(0, bases, offs)])


def test_wide_ragged_empty_and_short_reads(mcx, orc):
    bases, offs = synth.reads(3000, 130, genome_len=8000, seed=5, n_frac=0.3, var_len=True)
    for k in (65, 95, 97, 127):
        _compare(mcx, orc, k, 1, [(0, bases, offs)])
    e = np.zeros(0, np.uint8)
    _compare(mcx, orc, 95, 1, [(0, e, np.zeros(1, np.uint64))])
    b = np.frombuffer(b"ACGT" * 20, np.uint8)  # every read shorter than k
    _compare(mcx, orc, 127, 1, [(0, np.tile(b, 4), np.arange(0, 5 * 80, 80, dtype=np.uint64)[:5])])


def test_wide_colours_and_quality_cutoffs(mcx, orc):
    g0 = synth.genome(20000, seed=9)
    jobs = []
    for c in range(3):
        bases, offs = synth.reads(800, 250, seed=20 + c, g=g0, n_frac=0.02)
        jobs.append((c, bases, offs))
    jobs.append((1, *synth.reads(300, 250, seed=31, g=g0)))
    _compare(mcx, orc, 95, 3, jobs, names=["alice", "bob", "carol"])
    _compare(mcx, orc, 127, 3, jobs, names=["alice", "bob", "carol"])
    bases, offs = synth.reads(600, 260, genome_len=15000, seed=12, n_frac=0.02)
    rng = np.random.default_rng(3)
    quals = rng.integers(33, 74, len(bases)).astype(np.uint8)
    hp = bases.copy()
    hp[1000:1040] = ord("A")
    for fq, hc in ((0, 0), (40, 0), (0, 6), (38, 9)):
        _compare(mcx, orc, 77, 1, [(0, hp, offs, quals)], fq_cutoff=fq, hp_cutoff=hc)


def test_wide_stream_tile_seams_and_long_contigs(mcx, orc):
    """One long valid sequence over many tiles (every position of every lane holds a k-mer) and k-mers that straddle
    tile boundaries: the device stream path."""
    import torch
    g0 = synth.genome(3 * 4096 + 777, seed=4)
    s = bytes(g0)
    for k in (65, 95, 127):
        og = orc.Graph(k, 1, 1 << 18)
        b = np.frombuffer(s, np.uint8)
        st = og.add_reads(0, b, np.array([0, len(b)], np.uint64))
        want = og.ctx_bytes(True)[og.header_size():]
        g = mcx.Graph(k, 1, 1 << 18)
        t = torch.frombuffer(bytearray(s + b"\n" * 64), dtype=torch.uint8).cuda()
        g.add_stream_dev(0, t, len(s))
        g.sync()
        assert g.export(True) == want
        d = g.device_stats()
        assert (d.num_kmers_loaded, d.contigs_parsed) == (st.num_kmers_loaded, st.contigs_parsed) == (len(s) - k + 1, 1)
        # the packed form of the same stream
        g2 = mcx.Graph(k, 1, 1 << 18)
        g2.add_reads(0, b, np.array([0, len(b)], np.uint64))
        g2.sync()
        assert g2.export(True) == want and g2.checksum() == g.checksum()
        g.close(); g2.close()


def test_wide_records_checksum_load_and_sort(mcx, orc):
    """The .ctx records of a k = 95 / 127 graph: order-independent checksum (table scan == records), loading them into
    a fresh graph (`--graph`), and the `sort` / `index` device primitives on shuffled records."""
    for k in (95, 127):
        bases, offs = synth.reads(1200, 280, genome_len=25000, seed=40 + k, n_frac=0.03)
        g = mcx.Graph(k, 1, 1 << 18)
        g.add_reads(0, bases, offs)
        g.sync()
        body = g.export(True)
        rs = 8 * g.W + 5
        n = len(body) // rs
        assert n == g.nkmers and n > 1000
        cs, cnt = g.checksum()
        assert cnt == n and cs == mcx.records_checksum(body, k, 1)
        assert mcx.records_sorted(body, k, 1) == -1
        recs = np.frombuffer(body, np.uint8).reshape(n, rs)
        perm = np.random.default_rng(k).permutation(n)
        shuffled = np.ascontiguousarray(recs[perm])
        assert mcx.records_sorted(shuffled.tobytes(), k, 1) >= 0
        assert bytes(mcx.sort_records(shuffled.tobytes(), k, 1)) == body
        h = mcx.Graph(k, 1, 1 << 18)
        h.add_records(shuffled.tobytes(), 1, [(0, 0)])
        h.sync()
        assert h.export(True) == body and h.checksum() == (cs, cnt)
        g.close(); h.close()


def test_wide_refusals(mcx):
    """What k > 63 does not have: the partitioned insert (silently the fused kernel), several devices, shards."""
    g = mcx.Graph(95, 1, 1 << 18)
    g.configure("defer", 1)  # accepted, ignored
    bases, offs = synth.reads(200, 200, genome_len=5000, seed=1)
    g.add_reads(0, bases, offs)
    g.sync()
    assert g.nkmers > 0 and g.insert_stats()["flushes"] == 0
    g.close()
    with pytest.raises(mcx.McxError):
        mcx.Graph(95, 1, 1 << 20, devices=[0, 0])
    with pytest.raises(mcx.McxError):
        mcx.Graph(129, 1, 1 << 18)


def _run(maxk, *args):
    p = subprocess.run([os.path.join(BIN, "mccortex%d" % maxk)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stderr.decode(errors="replace")


@pytest.mark.parametrize("maxk,k", [(95, 65), (95, 95), (127, 97), (127, 127)])
def test_wide_cli_build_matches_oracle_ctx(mcx, orc, tmp_path, maxk, k):
    """`mccortex95 build` / `mccortex127 build` (the reference's MAXK builds): the whole file, header included."""
    bases, offs = synth.reads(900, 260, genome_len=20000, seed=k, n_frac=0.04)
    fq = tmp_path / "r.fq"
    with open(fq, "wb") as f:
        for i in range(len(offs) - 1):
            r = bytes(bases[int(offs[i]):int(offs[i + 1])])
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    out = tmp_path / "o.ctx"
    rc, err = _run(maxk, "build", "-k", str(k), "-n", "256K", "--sort", "-s", "wide", "--seq", str(fq), str(out))
    assert rc == 0, err
    og = orc.Graph(k, 1, 1 << 18)
    st = og.add_reads(0, bases, offs)
    og.update_stats(0, st)
    og.set_sample(0, "wide")
    assert out.read_bytes() == og.ctx_bytes(True)
    rc, err = _run(maxk, "index", str(out))
    assert rc == 0, err
    rc, err = _run(maxk, "build", "-k", str(k - 64 if maxk == 127 else 63), "-s", "a", "--seq", str(fq), str(tmp_path / "x.ctx"))
    assert rc == 1 and "Please recompile" in err
