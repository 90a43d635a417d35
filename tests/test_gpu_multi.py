"""One table over several GPUs behind the C ABI (mcx_graph_create_multi, csrc/mcx_multi.h): the
facade handle against the oracle, bit-exact.  The test box has one GPU: the same device is named
two or four times, which exercises everything but the xGMI hop itself (the peer copy becomes a
device-local copy).  Reference call site replaced: the batch loop of ctx_build.c:384-407."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _multi(mcx, k, ncols, devices, cap=1 << 20):
    g = mcx.Graph(k, ncols, cap, devices=devices)
    assert g.ndevices == len(devices)
    return g


@pytest.mark.parametrize("k,ndev,xch", [(31, 2, "v3"), (31, 4, "v3"), (31, 8, "v3"), (63, 2, "v3"), (63, 4, "v2"), (31, 2, "v2"), (31, 4, "v2"), (21, 2, "v2")])
def test_multi_add_reads_matches_oracle(mcx, orc, k, ndev, xch, monkeypatch):
    """both exchange formats of the in-process table: v3 (minimizer-owned super-k-mer records, ordinary
    per-shard tables; odd k >= 29) and v2 (hash-prefix shards, packed occurrences; any k)"""
    monkeypatch.setenv("MCX_MULTI_EXCHANGE", xch)
    bases, offs = synth.reads(6000, 120, genome_len=60000, seed=k + ndev, n_frac=0.05, lower_frac=0.1)
    og = orc.Graph(k, 1, 1 << 20)
    ost = og.add_reads(0, bases, offs)
    g = _multi(mcx, k, 1, [0] * ndev)
    g.add_reads(0, bases, offs)
    g.sync()
    st = g.device_stats()
    for f in ("num_good_reads", "num_bad_reads", "total_bases_loaded", "contigs_parsed", "num_kmers_loaded", "num_kmers_novel"):
        assert getattr(st, f) == getattr(ost, f), f
    assert g.nkmers == og.nkmers
    want = og.ctx_bytes(True)[og.header_size():]
    assert g.export(True) == want
    rs = 8 * g.W + 5
    a = np.frombuffer(g.export(False), np.uint8).reshape(-1, rs)
    b = np.frombuffer(want, np.uint8).reshape(-1, rs)
    assert sorted(map(bytes, a)) == sorted(map(bytes, b))
    cs, n = g.checksum()
    assert n == og.nkmers and cs == mcx.records_checksum(want, k, 1)
    # which shard holds a key is what mcx_graph_key_owner says (minimizer hash in v3, hash prefix in v2)
    rs = 8 * g.W + 5
    recs = np.frombuffer(want, np.uint8).reshape(-1, rs)
    for row in recs[:: max(1, len(recs) // 50)]:
        kw = [int(x) for x in np.frombuffer(row[:8 * g.W].tobytes(), "<u8")]
        assert 0 <= g.key_owner(kw) < ndev
    g.close()


def test_multi_small_pieces_and_colours(mcx, orc, monkeypatch):
    """many exchange pieces per batch (double buffering, slot reuse) and two colours"""
    monkeypatch.setenv("MCX_MULTI_PIECE", "20000")
    k = 31
    b0, o0 = synth.reads(4000, 100, genome_len=30000, seed=3)
    b1, o1 = synth.reads(3000, 100, genome_len=30000, seed=3, err=0.01)
    og = orc.Graph(k, 2, 1 << 20)
    og.add_reads(0, b0, o0)
    og.add_reads(1, b1, o1)
    og.add_reads(0, b1, o1)
    g = _multi(mcx, k, 2, [0, 0])
    g.add_reads(0, b0, o0)
    g.add_reads(1, b1, o1)
    g.add_reads(0, b1, o1)
    assert g.nkmers == og.nkmers
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    nk, sc = g.kmer_covg()
    h = g.covg_histogram(16)
    assert sum(h) == og.nkmers and len(nk) == 2
    g.close()


def test_multi_quality_and_homopolymer_cutoffs(mcx, orc):
    k = 21
    rng = np.random.default_rng(9)
    bases, offs = synth.reads(3000, 90, genome_len=20000, seed=4, n_frac=0.03)
    quals = rng.integers(33, 74, len(bases)).astype(np.uint8)
    for fq, hp in [(33 + 12, 0), (0, 4), (33 + 8, 5)]:
        og = orc.Graph(k, 1, 1 << 20)
        ost = og.add_reads(0, bases, offs, quals=quals, fq_cutoff=fq, hp_cutoff=hp)
        g = _multi(mcx, k, 1, [0, 0])
        g.add_reads(0, bases, offs, quals=quals, fq_cutoff=fq, hp_cutoff=hp)
        st = g.device_stats()
        assert st.num_kmers_loaded == ost.num_kmers_loaded and st.contigs_parsed == ost.contigs_parsed
        assert st.num_good_reads == ost.num_good_reads and st.num_bad_reads == ost.num_bad_reads
        assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
        g.close()


@pytest.mark.parametrize("k,paired,matedir,nbatches", [(31, False, "FF", 1), (31, True, "FR", 3), (63, True, "RF", 2), (21, False, "RR", 2)])
def test_multi_remove_pcr(mcx, orc, k, paired, matedir, nbatches):
    """the duplicate filter with the read-start table spread over the shards (k_pcr_claim / k_pcr_decide_claims)"""
    n = 6000
    bases, offs = synth.reads(n, 90, genome_len=600, seed=k + nbatches, n_frac=0.1, lower_frac=0.1, var_len=(nbatches > 1))
    g = _multi(mcx, k, 1, [0, 0])
    og = orc.Graph(k, 1, 1 << 20)
    st, ot = mcx.LoadStats(), orc.Stats()
    odup = [0, 0, 0]
    cuts = [2 * (n // 2 * i // nbatches) for i in range(nbatches + 1)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = offs[lo:hi + 1]
        g.add_reads_pcr(0, bases, o, paired=paired, matedir=matedir, stats=st)
        _, d = og.add_reads_pcr(0, bases, o, paired=paired, matedir=matedir, stats=ot)
        odup = [x + y for x, y in zip(odup, d)]
    assert (st.num_dup_se_reads, st.num_dup_pe_pairs, st.num_pe_reads) == tuple(odup)
    assert odup[1 if paired else 0] > n // 20
    cur = g.device_stats()
    for f in ("num_good_reads", "num_bad_reads", "total_bases_loaded", "contigs_parsed", "num_kmers_loaded", "num_kmers_novel"):
        assert getattr(cur, f) == getattr(ot, f), f
    assert g.nkmers == og.nkmers
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.close()


def test_multi_add_records_and_stream(mcx, orc):
    import torch
    k = 31
    bases, offs = synth.reads(3000, 100, genome_len=20000, seed=21)
    og = orc.Graph(k, 1, 1 << 20)
    og.add_reads(0, bases, offs)
    body = og.ctx_bytes(True)[og.header_size():]
    g = _multi(mcx, k, 1, [0, 0])
    st = g.add_records(body, 1, [(0, 0)])
    assert st.nkmers_loaded == og.nkmers and st.nkmers_novel == og.nkmers
    # ... and the same reads once more as a device-resident stream: coverage doubles
    stream = torch.from_numpy(synth.to_stream(bases, offs)).cuda()
    g.add_stream_dev(0, stream, stream.numel())
    g.sync()
    og.add_reads(0, bases, offs)
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    with pytest.raises(mcx.McxError):
        g.configure("intersect", 1)
    g.close()


def test_multi_needs_power_of_two(mcx):
    with pytest.raises(mcx.McxError):
        mcx.Graph(31, 1, 1 << 20, devices=[0, 0, 0])


@pytest.mark.parametrize("k,xch", [(31, "v3"), (63, "v3"), (31, "v2"), (63, "v2")])
def test_multi_hot_kmers_spill_instead_of_failing(mcx, orc, k, xch, monkeypatch):
    """Low-complexity input on a sharded table: a 3 Mbase poly-G read and 4000 poly-A reads put
    millions of occurrences of ONE k-mer on one owner -- far beyond its (owner, region) segment
    (mean + 8 sigma) and the owner's overflow bin (>= 64 K tuples).  One GPU handles that with a direct
    insert on the spot; the multi-GPU sender spills what fits nowhere and the host routes it to the
    owners (mcx_multi.h, group_route_spill).  Same graph as the oracle, in one piece and in many."""
    monkeypatch.setenv("MCX_MULTI_EXCHANGE", xch)
    rng = np.random.default_rng(5)
    reads = [b"G" * 3_000_000]
    reads += [b"A" * 150] * 4000
    reads += [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 150)) for _ in range(2000)]
    reads += [b"ACACACACAC" * 30000]
    bases, offs = orc.pack_reads(reads)
    og = orc.Graph(k, 1, 1 << 20)
    ost = og.add_reads(0, bases, offs)
    want = og.ctx_bytes(True)[og.header_size():]
    for piece in (None, "300000"):
        if piece:
            os.environ["MCX_MULTI_PIECE"] = piece
        try:
            g = _multi(mcx, k, 1, [0, 0])
            g.add_reads(0, bases, offs)
            g.add_reads(0, bases, offs)  # the send sets are reused: what they spilled is routed first
            g.sync()
            st = g.device_stats()
            assert st.num_kmers_loaded == 2 * ost.num_kmers_loaded
            assert g.nkmers == og.nkmers
            body = g.export(True)
            g.close()
        finally:
            os.environ.pop("MCX_MULTI_PIECE", None)
        rs = 8 * ((2 * k + 63) // 64) + 5
        a = np.frombuffer(body, np.uint8).reshape(-1, rs).copy()
        b = np.frombuffer(want, np.uint8).reshape(-1, rs)
        assert (a[:, :rs - 5] == b[:, :rs - 5]).all() and (a[:, -1] == b[:, -1]).all()   # keys and edges
        ca = a[:, rs - 5:rs - 1].copy().view("<u4")[:, 0].astype(np.int64)
        cb = b[:, rs - 5:rs - 1].copy().view("<u4")[:, 0].astype(np.int64)
        assert (ca == np.minimum(2 * cb, 0xFFFFFFFF)).all()                               # every read twice


@pytest.mark.parametrize("k,ncols,ndev", [(31, 1, 4), (63, 2, 2), (21, 3, 8)])
def test_multi_sorted_export_merges_per_shard_runs(mcx, orc, k, ncols, ndev, monkeypatch):
    """`--sort` on a multi-GPU table: above MCX_MULTI_SORT_DEV records nothing is gathered on one device;
    every shard sorts its own records and the host merges the N sorted runs (hash_table.c:371 order:
    key words compared top word first).  Forced here for a small graph; same bytes as the oracle and
    as the one-device-sort path."""
    g0 = synth.genome(50000, 3)
    jobs = []
    for c in range(ncols):
        b, o = synth.reads(5000, 100, seed=40 + c, g=g0, n_frac=0.03, err=0.003)
        jobs.append((c, b, o))
    og = orc.Graph(k, ncols, 1 << 20)
    for c, b, o in jobs:
        og.add_reads(c, b, o)
    want = og.ctx_bytes(True)[og.header_size():]
    got = {}
    for lim in ("0", "1000000000"):
        monkeypatch.setenv("MCX_MULTI_SORT_DEV", lim)
        g = _multi(mcx, k, ncols, [0] * ndev)
        for c, b, o in jobs:
            g.add_reads(c, b, o)
        got[lim] = g.export(True)
        g.close()
    assert got["0"] == want and got["1000000000"] == want


@pytest.mark.parametrize("k", [31, 63])
def test_multi_v3_record_segments_spill(mcx, orc, k, monkeypatch):
    """Exchange format v3 with record segments far too small (MCX_MULTI_SKCAP): nearly every super-k-mer
    record takes the sender's spill area and is routed to its owner by the host (k_superk_pick); same
    graph as the oracle."""
    monkeypatch.setenv("MCX_MULTI_EXCHANGE", "v3")
    monkeypatch.setenv("MCX_MULTI_SKCAP", "64")
    monkeypatch.setenv("MCX_MULTI_PIECE", "400000")
    bases, offs = synth.reads(9000, 130, genome_len=80000, seed=k, n_frac=0.04, lower_frac=0.05)
    og = orc.Graph(k, 1, 1 << 20)
    ost = og.add_reads(0, bases, offs)
    g = _multi(mcx, k, 1, [0, 0, 0, 0])
    g.add_reads(0, bases, offs)
    g.sync()
    st = g.device_stats()
    assert st.num_kmers_loaded == ost.num_kmers_loaded and st.contigs_parsed == ost.contigs_parsed
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.close()


def _owner_of_read_kmers(mcx, read, k, nparts):
    owners = set()
    for i in range(len(read) - k + 1):
        key, _ = mcx.kmer_canonical(mcx.kmer_from_str(read[i:i + k], k), k)
        owners.add(mcx.superk_owner(key, k, nparts))
    return owners


@pytest.mark.parametrize("k", [31, 63])
def test_multi_v3_every_record_spills_and_one_owner_takes_all(mcx, orc, k, monkeypatch):
    """The two ends of exchange v3's record budget (round 5: no input can end in MCX_ERR_FULL where one GPU succeeds).
    (a) Reads of k .. k + 2 bases: every record holds 1-3 k-mers -- the most records per k-mer an input makes -- with
    segments of 16 records (MCX_MULTI_SKCAP), so nearly all of them go through the sender's spill area, which now
    holds one record per start position of a piece (a record is a run of >= 1 k-mers: it cannot overflow).
    (b) Reads chosen so that ONE shard owns every k-mer (by the product's own minimizer-owner function): the owner
    side books a true upper bound and settles it, nothing falls back to the per-occurrence insert, and the other
    shards stay empty."""
    monkeypatch.setenv("MCX_MULTI_EXCHANGE", "v3")
    rng = np.random.default_rng(k)
    gen = synth.genome(40000, 5)
    # (a)
    monkeypatch.setenv("MCX_MULTI_PIECE", "400000")
    monkeypatch.setenv("MCX_MULTI_SKCAP", "16")
    reads = []
    for _ in range(12000):
        n = int(rng.integers(k, k + 3))
        p = int(rng.integers(0, len(gen) - n))
        reads.append(bytes(gen[p:p + n]))
    bases, offs = orc.pack_reads(reads)
    og = orc.Graph(k, 1, 1 << 20)
    ost = og.add_reads(0, bases, offs)
    g = _multi(mcx, k, 1, [0] * 8)
    g.add_reads(0, bases, offs)
    g.sync()
    st = g.device_stats()
    assert st.num_kmers_loaded == ost.num_kmers_loaded and st.contigs_parsed == ost.contigs_parsed
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    ist = g.insert_stats()
    assert ist["spilled"] > len(reads) // 4, ist   # (a read is at least one record; 8 x 8 x 16 segment slots per piece)
    g.close()
    monkeypatch.delenv("MCX_MULTI_SKCAP")
    monkeypatch.setenv("MCX_MULTI_PIECE", "30000")
    # (b)
    nparts = 4
    mine = []
    while len(mine) < 1500:
        n = int(rng.integers(k, k + 4))
        p = int(rng.integers(0, len(gen) - n))
        r = bytes(gen[p:p + n])
        if _owner_of_read_kmers(mcx, r.decode(), k, nparts) == {2}:
            mine.append(r)
    bases, offs = orc.pack_reads(mine)
    og = orc.Graph(k, 1, 1 << 20)
    g = _multi(mcx, k, 1, [0] * nparts)
    for _ in range(3):
        g.add_reads(0, bases, offs)
        og.add_reads(0, bases, offs)
    g.sync()
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    ist = g.insert_stats()
    assert ist["fallback_inserts"] == 0 and ist["foreign_inserts"] == 0, ist
    g.close()


def test_insert_stats_make_the_slow_paths_visible(mcx, orc):
    """mcx_graph_insert_stats: zero on well-spread input; hot k-mers (thousands of occurrences of one key in a batch)
    overflow their partition bin and are counted as fallback inserts; keys of another shard handed to a shard handle
    are counted as foreign inserts -- both correct, both slow, neither silent any more."""
    k = 31
    bases, offs = synth.reads(20000, 150, genome_len=300000, seed=9)
    g = mcx.Graph(k, 1, 1 << 22)
    g.add_reads(0, bases, offs)
    ist = g.insert_stats()
    assert ist["fallback_inserts"] == 0 and ist["foreign_inserts"] == 0 and ist["spilled"] == 0 and ist["flushes"] >= 1, ist
    g.close()
    hot, hoffs = orc.pack_reads(["A" * 150] * 200000)
    g = mcx.Graph(k, 1, 1 << 22)
    og = orc.Graph(k, 1, 1 << 22)
    g.add_reads(0, hot, hoffs)
    og.add_reads(0, hot, hoffs)
    ist = g.insert_stats()
    assert ist["fallback_inserts"] > 0, ist
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.reset()
    assert g.insert_stats()["fallback_inserts"] == 0
    g.close()
    sh = mcx.Graph(k, 1, 1 << 22, nparts=2, part=0)
    sh.add_reads(0, bases, offs)
    ist = sh.insert_stats()
    assert ist["foreign_inserts"] > 0, ist
    sh.close()
