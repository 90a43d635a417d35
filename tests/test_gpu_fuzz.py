"""Randomised parity: arbitrary byte streams, every odd k, ragged stream lengths."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _oracle_body(orc, k, stream_bytes, ncap=1 << 20):
    og = orc.Graph(k, 1, ncap)
    b = np.frombuffer(stream_bytes, np.uint8)
    st = og.add_reads(0, b, np.array([0, len(b)], np.uint64))  # one "read": any non-ACGT byte splits it
    return og, st, og.ctx_bytes(True)[og.header_size():]


def _gpu_stream(mcx, k, stream_bytes, defer, ncap=1 << 20):
    import torch
    g = mcx.Graph(k, 1, ncap)
    g.configure("defer", defer)
    n = len(stream_bytes)
    t = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")  # slack after the stream is never read as data
    t[n:] = ord("A")                                            # ... even if it looks like bases
    if n:
        t[:n] = torch.frombuffer(bytearray(stream_bytes), dtype=torch.uint8).cuda()
    g.add_stream_dev(0, t, n)
    g.sync()
    return g


@pytest.mark.parametrize("k", list(range(3, 64, 2)))
def test_every_odd_k(mcx, orc, k):
    rng = np.random.default_rng(k)
    s = bytes(rng.choice(np.frombuffer(b"ACGTACGTACGTacgtN\n", np.uint8), 6000))
    og, st, want = _oracle_body(orc, k, s)
    g = _gpu_stream(mcx, k, s, defer=k % 4 == 1)
    assert g.export(True) == want
    d = g.device_stats()
    assert (d.num_kmers_loaded, d.contigs_parsed, d.total_bases_loaded) == (st.num_kmers_loaded, st.contigs_parsed, st.total_bases_loaded)
    g.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MCX_FUZZ_SEEDS_BYTES", "6"))))
def test_arbitrary_bytes_and_ragged_lengths(mcx, orc, seed):
    rng = np.random.default_rng(100 + seed)
    alphabet = np.concatenate([np.frombuffer(b"ACGT" * 40 + b"acgtNn\n\r\t @+>", np.uint8), rng.integers(0, 256, 30).astype(np.uint8)])
    for n in (0, 1, 30, 31, 32, 4095, 4096, 4097, 4096 + 15, 8192 + 33, 50001):
        s = bytes(rng.choice(alphabet, n)) if n else b""
        for k in (31, 63, 5):
            og, st, want = _oracle_body(orc, k, s)
            for defer in (0, 1):
                g = _gpu_stream(mcx, k, s, defer)
                assert g.export(True) == want, (n, k, defer)
                assert g.device_stats().num_kmers_loaded == st.num_kmers_loaded
                g.close()


def test_stream_boundaries_are_separators(mcx, orc):
    """A k-mer must never span two add_stream_dev calls, nor read beyond nbytes."""
    import torch
    a, b = b"ACGT" * 20, b"TTGCA" * 20
    og = orc.Graph(31, 1, 1 << 16)
    bases, offs = orc.pack_reads([a, b])
    og.add_reads(0, bases, offs)
    want = og.ctx_bytes(True)[og.header_size():]
    buf = torch.frombuffer(bytearray(a + b + b"ACGT" * 16), dtype=torch.uint8).cuda()
    g = mcx.Graph(31, 1, 1 << 16)
    pad = torch.zeros(16 * 20, dtype=torch.uint8, device="cuda")
    pad[:len(b) + 64] = buf[len(a):len(a) + len(b) + 64]   # 16-byte aligned copy of the second read + trailing bases
    g.add_stream_dev(0, buf, len(a))
    g.add_stream_dev(0, pad, len(b))
    assert g.export(True) == want
    g.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MCX_FUZZ_SEEDS_BYTES", "4"))))
def test_superkmer_records_on_arbitrary_bytes(mcx, orc, seed):
    """Exchange format v3 on arbitrary byte streams and ragged lengths: the union of the owner
    tables (simulated shards) is the oracle's graph."""
    import torch
    from mccortex_amd import shard
    rng = np.random.default_rng(500 + seed)
    alphabet = np.concatenate([np.frombuffer(b"ACGT" * 60 + b"acgtNn\n\r @+>", np.uint8), rng.integers(0, 256, 10).astype(np.uint8)])
    for n in (0, 1, 29, 31, 32, 47, 4095, 4096, 4097, 4096 * 3 + 7, 60001):
        s = bytes(rng.choice(alphabet, n)) if n else b""
        for k, nparts in ((31, 4), (29, 2), (31, 32), (63, 4), (35, 8)):
            og, st, want = _oracle_body(orc, k, s)
            graphs = [mcx.Graph(k, 1, 1 << 20) for _ in range(nparts)]
            t = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            t[n:] = ord("A")
            if n:
                t[:n] = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
            segs, cap = graphs[0].superk_layout(nparts, max(n, 1))
            W = (2 * k + 63) // 64
            recs = torch.zeros((nparts, segs, cap, 2 * W), dtype=torch.int64, device="cuda")
            fills = torch.zeros((segs, nparts), dtype=torch.int64, device="cuda")
            graphs[0].superk_bins_dev(t, n, nparts, recs, fills, cap)
            graphs[0].sync()
            assert graphs[0].device_stats().num_kmers_loaded == st.num_kmers_loaded, (n, k, nparts)
            counts = fills.t().contiguous()
            bodies = []
            for o, g in enumerate(graphs):
                g.add_superk_dev(0, recs[o], counts[o], segs, cap, int(counts[o].sum()) * 16)
                bodies.append(g.export(True))
                g.close()
            assert shard.merge_sorted_bodies(bodies, 8 * W + 5, 8 * W) == want, (n, k, nparts)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MCX_FUZZ_SEED0", "0")), int(os.environ.get("MCX_FUZZ_SEED0", "0")) + int(os.environ.get("MCX_FUZZ_SEEDS", "6"))))
def test_random_colours_devices_and_batches(mcx, orc, seed, monkeypatch):
    """Random k, colours, device count, exchange format, pool of L1 bin sets, flush size and batch cuts;
    samples in random order (colour switches at every batch); arbitrary bytes in some reads.  The graph
    must be the oracle's whatever the shape (MCX_FUZZ_SEEDS widens the run for a soak)."""
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([5, 11, 21, 27, 29, 31, 33, 41, 55, 63]))
    ncols = int(rng.integers(1, 5))
    ndev = int(rng.choice([1, 1, 2, 4, 8]))
    monkeypatch.setenv("MCX_MULTI_EXCHANGE", str(rng.choice(["v2", "v3"])))
    monkeypatch.setenv("MCX_L1_SETS", str(int(rng.choice([1, 2, 5, 32]))))
    monkeypatch.setenv("MCX_MULTI_PIECE", str(int(rng.choice([40000, 250000, 1 << 27]))))
    g0 = synth.genome(int(rng.integers(3000, 60000)), seed)
    alphabet = np.frombuffer(b"ACGT" * 30 + b"acgtNn-*", np.uint8)
    jobs = []
    for _ in range(int(rng.integers(3, 14))):
        n = int(rng.integers(1, 2500))
        ln = int(rng.integers(1, 260))
        b, o = synth.reads(n, ln, seed=int(rng.integers(1 << 30)), g=g0, n_frac=float(rng.choice([0, 0.05, 0.5])), lower_frac=0.1, err=0.003)
        if rng.random() < 0.3:  # a few reads of arbitrary bytes, ragged lengths (among them empty ones)
            extra = [bytes(rng.choice(alphabet, int(rng.integers(0, 400)))) for _ in range(int(rng.integers(1, 40)))]
            b2, o2 = orc.pack_reads(extra)
            b = np.concatenate([b, b2]); o = np.concatenate([o, o[-1] + o2[1:]])
        jobs.append((int(rng.integers(0, ncols)), b, o))
    og = orc.Graph(k, ncols, 1 << 20)
    tot = 0
    for c, b, o in jobs:
        tot += og.add_reads(c, b, o).num_kmers_loaded
    want = og.ctx_bytes(True)[og.header_size():]
    g = mcx.Graph(k, ncols, 1 << 20, devices=[0] * ndev) if ndev > 1 else mcx.Graph(k, ncols, 1 << 20)
    g.configure("defer_tuples", int(rng.choice([1 << 20, 1 << 22, 1 << 26])))
    for c, b, o in jobs:
        g.add_reads(c, b, o)
    assert g.device_stats().num_kmers_loaded == tot, (seed, k, ncols, ndev)
    assert g.nkmers == og.nkmers, (seed, k, ncols, ndev)
    assert g.export(True) == want, (seed, k, ncols, ndev)
    g.close()
