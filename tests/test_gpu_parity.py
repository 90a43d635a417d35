"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _oracle(orc, k, ncols, jobs, cap=1 << 20, **kw):
    """jobs: list of (colour, bases, offs[, quals]) 'files' in order."""
    og = orc.Graph(k, ncols, cap)
    stats = []
    for job in jobs:
        col, bases, offs = job[:3]
        quals = job[3] if len(job) > 3 else None
        st = og.add_reads(col, bases, offs, quals=quals, **kw)
        og.update_stats(col, st)
        stats.append(st)
    return og, stats


def _gpu(mcx, k, ncols, jobs, cap=1 << 20, **kw):
    g = mcx.Graph(k, ncols, cap)
    hdr = mcx.CtxHeader(k, ncols)
    stats = []
    prev = g.device_stats()
    for job in jobs:
        col, bases, offs = job[:3]
        quals = job[3] if len(job) > 3 else None
        st = g.add_reads(col, bases, offs, quals=quals, **kw)
        cur = g.device_stats()
        for f in ("num_good_reads", "num_bad_reads", "total_bases_loaded", "contigs_parsed",
                  "num_kmers_loaded", "num_kmers_novel"):
            setattr(st, f, getattr(cur, f) - getattr(prev, f))
        prev = cur
        hdr.update_stats(col, st.total_bases_loaded, st.contigs_parsed)
        stats.append(st)
    return g, hdr, stats


def _compare(mcx, orc, k, ncols, jobs, cap=1 << 20, names=None, **kw):
    og, ost = _oracle(orc, k, ncols, jobs, cap, **kw)
    g, hdr, gst = _gpu(mcx, k, ncols, jobs, cap, **kw)
    if names:
        for c, n in enumerate(names):
            og.set_sample(c, n)
            hdr.names[c] = n
    for a, b in zip(gst, ost):
        assert {f: v for f, v in a.as_dict().items() if f in b.as_dict()} == b.as_dict()  # (the duplicate counters are not the oracle's)
    assert g.nkmers == og.nkmers
    want = og.ctx_bytes(True)
    got = mcx.ctx_header_bytes(hdr) + g.export(True)
    assert len(got) == len(want)
    assert got == want
    # unsorted export holds the same record set
    body = g.export(False)
    rs = 8 * g.W + 5 * ncols
    a = np.frombuffer(body, np.uint8).reshape(-1, rs)
    b = np.frombuffer(want[og.header_size():], np.uint8).reshape(-1, rs)
    assert sorted(map(bytes, a)) == sorted(map(bytes, b))
    g.close()
    return og


@pytest.mark.parametrize("k", [31, 21, 3, 63, 39, 33])
def test_random_reads_match_oracle(mcx, orc, k):
    bases, offs = synth.reads(3000, 100, genome_len=20000, seed=k, n_frac=0.05, lower_frac=0.1)
    _compare(mcx, orc, k, 1, [(0, bases, offs)])


@pytest.mark.parametrize("rep", [1, 16, 32, 64])
def test_replicas_of_the_region_bins(mcx, orc, rep, monkeypatch):
    """The region bins exist in `rep` replicas (8, or 32 for a large one-colour window: mcx_api.hip rep1); MCX_REP1
    forces the count, so the small graphs of this suite meet the other layouts too."""
    monkeypatch.setenv("MCX_REP1", str(rep))
    bases, offs = synth.reads(4000, 120, genome_len=30000, seed=70 + rep, n_frac=0.05, lower_frac=0.1)
    _compare(mcx, orc, 31, 1, [(0, bases, offs)])
    _compare(mcx, orc, 55, 2, [(0, bases, offs), (1, bases[: int(offs[1500])], offs[:1501])], names=["a", "b"])


def test_placed_sub_table_bins(mcx, orc, monkeypatch):
    """place_bins: the sub-table bins are the best of n allocations by the write-pattern probe (one allocation for a
    small table; at the benchmark geometry -- tests/test_gpu_fullsize.py -- one per half of the flush overlap)."""
    monkeypatch.setenv("MCX_PLACE_BINS", "3")
    bases, offs = synth.reads(4000, 120, genome_len=30000, seed=91, n_frac=0.05)
    _compare(mcx, orc, 31, 1, [(0, bases, offs)])
    _compare(mcx, orc, 63, 1, [(0, bases, offs)], cap=1 << 22)


def test_ragged_and_empty_reads(mcx, orc):
    bases, offs = synth.reads(5000, 40, genome_len=5000, seed=3, n_frac=0.2, var_len=True)
    _compare(mcx, orc, 31, 1, [(0, bases, offs)])
    _compare(mcx, orc, 63, 1, [(0, bases, offs)])


def test_no_reads_and_all_short(mcx, orc):
    e = np.zeros(0, np.uint8)
    _compare(mcx, orc, 31, 1, [(0, e, np.zeros(1, np.uint64))])
    bases, offs = orc.pack_reads(["ACGT", "", "ACGTACGTAC", "NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN"])
    _compare(mcx, orc, 31, 1, [(0, bases, offs)])


def test_reference_build_kat(mcx, orc):
    # the reads the reference's own unit test loads (src/tests/build_graph_tests.c:41-121),
    # minus the pairs its PCR-duplicate filter drops; expected coverages 3/3 (:119-120)
    reads = ["CTACGATGTATGCTTAGCTGTTCCG", "TAGAACGTTCCCTACACGTCCTATG", "CTACGATGTATGCTTAGCTAATGAT",
             "TAGAACGTTCCCTACACGTTGTTTG", "ACGTGTAGGGAACGTTCTACTTCTACCGGAGGAT",
             "AGCTAAGCATACATCGTAGTACAATGCACCCTCC"]
    bases, offs = orc.pack_reads(reads)
    og = _compare(mcx, orc, 19, 1, [(0, bases, offs)], cap=1024)
    assert og.lookup("CTACGATGTATGCTTAGCT")[0][0] == 3
    assert og.lookup("TAGAACGTTCCCTACACGT")[0][0] == 3


def test_multi_colour_and_files(mcx, orc):
    g = synth.genome(30000, 5)
    jobs = []
    for c in range(3):
        for f in range(2):
            b, o = synth.reads(1500, 120, seed=10 * c + f, g=g, lower_frac=0.05)
            jobs.append((c, b, o))
    _compare(mcx, orc, 31, 3, jobs, names=["alice", "bob", "carol"])
    _compare(mcx, orc, 51, 3, jobs, names=["alice", "bob", "carol"])


def test_hot_keys_contention(mcx, orc):
    # homopolymer and short-period reads: thousands of occurrences of a handful of keys
    reads = ["A" * 150] * 3000 + ["ACACACACACACACACACACACACACACACACACACACACACACACACAC" * 3] * 2000 + ["T" * 200] * 500
    bases, offs = orc.pack_reads(reads)
    _compare(mcx, orc, 31, 1, [(0, bases, offs)])
    _compare(mcx, orc, 63, 1, [(0, bases, offs)])


def test_quality_and_homopolymer_cutoffs(mcx, orc):
    bases, offs = synth.reads(2000, 120, genome_len=20000, seed=11, n_frac=0.05)
    rng = np.random.default_rng(5)
    quals = rng.integers(33, 74, len(bases)).astype(np.uint8)
    # plant homopolymer runs
    for p in rng.integers(0, len(bases) - 20, 300):
        bases[p:p + int(rng.integers(3, 15))] = ord("ACGT"[int(rng.integers(0, 4))])
    for fq, hp in [(33 + 10, 0), (0, 5), (33 + 5, 7), (33 + 20, 3)]:
        _compare(mcx, orc, 21, 1, [(0, bases, offs, quals)], fq_cutoff=fq, hp_cutoff=hp)
        _compare(mcx, orc, 41, 1, [(0, bases, offs, quals)], fq_cutoff=fq, hp_cutoff=hp)


def test_table_full_is_reported(mcx):
    bases, offs = synth.reads(4000, 100, genome_len=200000, seed=2)
    g = mcx.Graph(31, 1, 1024)
    g.add_reads(0, bases, offs)
    with pytest.raises(mcx.McxError) as ei:
        g.sync()
    assert ei.value.code == mcx.MCX_ERR_FULL and "Hash table is full" in str(ei.value)
    g.close()


def test_device_stream_and_sharded_path(mcx, orc):
    import torch
    k = 31
    for k in (31, 55):
        bases, offs = synth.reads(20000, 150, genome_len=100000, seed=k)
        og, _ = _oracle(orc, k, 1, [(0, bases, offs)])
        want = og.ctx_bytes(True)[og.header_size():]
        stream = torch.from_numpy(synth.to_stream(bases, offs)).cuda()
        # (a) fused device-stream entry
        g = mcx.Graph(k, 1, 1 << 21)
        g.add_stream_dev(0, stream, stream.numel())
        g.sync()
        assert g.export(True) == want
        # (b) partition into 4 owner bins, then insert each bin into its own shard graph
        W = g.W
        nparts, cap = 4, 1 << 20
        keys = torch.zeros((nparts, cap, W), dtype=torch.int64, device="cuda")
        edges = torch.zeros((nparts, cap), dtype=torch.uint8, device="cuda")
        counts = torch.zeros(nparts, dtype=torch.int64, device="cuda")
        g.partition_stream_dev(stream, stream.numel(), nparts, cap, keys, edges, counts)
        g.sync()
        torch.cuda.synchronize()
        cnt = counts.cpu().numpy()
        assert cnt.sum() == g.device_stats().num_kmers_loaded // 2  # counted by both launches
        bodies = []
        for p in range(nparts):
            sg = mcx.Graph(k, 1, 1 << 20)
            sg.insert_tuples_dev(0, keys[p], edges[p], int(cnt[p]))
            sg.sync()
            kk, cc, ee = sg.records(True)
            for row in kk[:50]:
                assert mcx.key_owner([int(x) for x in row], k, nparts) == p
            bodies.append(sg.export(True))
            sg.close()
        rs = 8 * W + 5
        allrec = np.concatenate([np.frombuffer(b, np.uint8).reshape(-1, rs) for b in bodies])
        wantrec = np.frombuffer(want, np.uint8).reshape(-1, rs)
        assert sorted(map(bytes, allrec)) == sorted(map(bytes, wantrec))
        g.close()


def test_config1_100k_x_100bp(mcx, orc):
    # BASELINE.json configs[0]: 100k x 100bp, k=31, 1 colour, bit-identical sorted .ctx
    g = synth.genome(1_000_000, 42)
    bases, offs = synth.reads(100_000, 100, seed=42, g=g)
    og = _compare(mcx, orc, 31, 1, [(0, bases, offs)], cap=4 << 20, names=["sample0"])
    assert og.nkmers > 1_000_000


@pytest.mark.parametrize("k,ncols", [(31, 1), (31, 3), (63, 1), (45, 2)])
def test_direct_and_deferred_paths_agree(mcx, orc, k, ncols):
    """The two insert strategies (HBM atomics vs partition + LDS insert) build the same graph,
    including multiple flushes, bin overflow -> direct-insert fallback (tiny bins, hot keys) and
    colour switches."""
    g0 = synth.genome(30000, 9)
    jobs = []
    for c in range(ncols):
        b, o = synth.reads(4000, 120, seed=50 + c, g=g0, n_frac=0.05, lower_frac=0.05)
        jobs.append((c, b, o))
    hb, ho = orc.pack_reads(["A" * 150] * 800 + ["ACGT" * 40] * 500)
    jobs.append((0, hb, ho))
    og, _ = _oracle(orc, k, ncols, jobs)
    want = og.ctx_bytes(True)[og.header_size():]
    for cfg in [{"defer": 0}, {"defer": 1}, {"defer": 1, "defer_tuples": 20000}, {"defer": 1, "defer_tuples": 300000},
                {"defer": 1, "flush_regions": 1}, {"defer": 1, "defer_tuples": 300000, "flush_regions": 3},
                {"defer": 1, "flush_regions": 100000}]:
        g = mcx.Graph(k, ncols, 1 << 20)
        for key, v in cfg.items():
            g.configure(key, v)
        for col, b, o in jobs:
            g.add_reads(col, b, o)
        assert g.nkmers == og.nkmers, cfg
        assert g.export(True) == want, cfg
        st = g.device_stats()
        assert st.num_kmers_novel == og.nkmers
        g.close()


def test_deferred_large_table_many_subtables(mcx, orc):
    """Enough slots for several L1 bins (nsub > 512) so both partition levels are exercised."""
    bases, offs = synth.reads(60000, 150, genome_len=400000, seed=21)
    og, _ = _oracle(orc, 31, 1, [(0, bases, offs)], cap=1 << 22)
    want = og.ctx_bytes(True)[og.header_size():]
    for regions in (0, 1, 5, 1 << 20):  # regions split + applied per flush step (0 = automatic)
        g = mcx.Graph(31, 1, 3 << 20)   # 768 sub-tables in 256 regions of 3
        g.configure("flush_regions", regions)
        g.configure("profile", 1)
        g.add_reads(0, bases, offs)
        assert g.export(True) == want, regions
        prof = g.profile()
        assert "k_stream_bin" in prof and "k_tuples_bin" in prof and "k_lds_insert" in prof
        if regions == 1:
            assert prof["k_lds_insert"][0] == 512 and prof["k_tuples_bin"][0] == 512  # one step per region
        g.close()


def test_configure_profile_and_device_memory(mcx):
    import ctypes as C
    free, total = C.c_uint64(), C.c_uint64()
    assert mcx.lib().mcx_device_memory(0, C.byref(free), C.byref(total)) == 0
    assert 0 < free.value <= total.value and total.value > (100 << 30)
    g = mcx.Graph(31, 1, 1 << 16)
    with pytest.raises(mcx.McxError):
        g.configure("no_such_key", 1)
    g.configure("profile", 1)
    b, o = synth.reads(500, 100, genome_len=5000, seed=1)
    g.add_reads(0, b, o)
    n1 = g.nkmers
    prof = g.profile()
    assert prof and all(c >= 1 and ms >= 0 for c, ms in prof.values())
    g.reset()
    assert g.nkmers == 0 and g.device_stats().num_kmers_loaded == 0
    g.configure("defer", 0)
    g.add_reads(0, b, o)
    assert g.nkmers == n1
    g.close()


@pytest.mark.parametrize("k,nparts", [(31, 4), (63, 2), (31, 8)])
def test_sharded_table_compact_exchange(mcx, orc, k, nparts):
    """N shards of one hash-prefix-sharded table, simulated on one GPU: every 'rank' k-merises its
    own reads into per-(owner, region) bins of packed tuples (exchange format v2), every owner
    consumes the blocks addressed to it; the union of the shard exports is the oracle's graph, every
    key sits on the shard mcx_graph_key_owner names, and tiny segments exercise the overflow bins."""
    import torch
    from mccortex_amd import shard
    g0 = synth.genome(60000, 31)
    cap = 1 << 20
    graphs = [mcx.Graph(k, 1, cap, nparts=nparts, part=p) for p in range(nparts)]
    W = graphs[0].W
    all_b, all_o = [], []
    for r in range(nparts):
        bases, offs = synth.reads(3000, 140, seed=70 + r, g=g0, n_frac=0.05)
        if r == 0:  # hot keys
            hb, ho = orc.pack_reads(["A" * 150] * 300 + ["ACGTTGCA" * 20] * 200)
            bases = np.concatenate([bases, hb]); offs = np.concatenate([offs, offs[-1] + ho[1:]])
        all_b.append(bases); all_o.append(offs)
        stream = torch.from_numpy(synth.to_stream(bases, offs)).cuda()
        ntup = 3500 * 120
        segs, seg_cap, ov_cap = graphs[r].shard_layout(ntup)
        if r == 1:
            seg_cap = 16  # almost everything overflows
            ov_cap = 1 << 20
        blk = shard.BlockExchange(nparts, segs, seg_cap, ov_cap, W, "cuda")
        blk.fill(graphs[r], stream, stream.numel())
        torch.cuda.synchronize(); graphs[r].sync()
        assert not blk.overflowed()
        if r == 1:
            assert int(blk.ov_counts.sum()) > 10000
        for o in range(nparts):  # "exchange": owner o takes block o of every sender
            rb = shard.BlockExchange(1, segs, seg_cap, ov_cap, W, "cuda")
            rb.keys[0], rb.counts[0], rb.ov_keys[0], rb.ov_edges[0], rb.ov_counts[0] = \
                blk.keys[o], blk.counts[o], blk.ov_keys[o], blk.ov_edges[o], blk.ov_counts[o]
            rb.consume(graphs[o], 0, ntup // nparts)
            graphs[o].sync()
    og = orc.Graph(k, 1, 1 << 21)
    ost = orc.Stats()
    for b, o in zip(all_b, all_o):
        og.add_reads(0, b, o, stats=ost)
    tot = [g.device_stats() for g in graphs]   # (v2 counts occurrences and contigs where the reads are k-merised)
    for f in ("num_kmers_loaded", "contigs_parsed", "total_bases_loaded", "num_kmers_novel"):
        assert sum(getattr(t, f) for t in tot) == getattr(ost, f), f
    want = og.ctx_bytes(True)[og.header_size():]
    bodies = []
    for p, g in enumerate(graphs):
        kk, cc, ee = g.records(True)
        for row in kk[::37]:
            assert g.key_owner([int(x) for x in row]) == p
        bodies.append(g.export(True))
        g.close()
    assert shard.merge_sorted_bodies(bodies, 8 * W + 5, 8 * W) == want
    assert all(len(b) > 0 for b in bodies)


@pytest.mark.parametrize("k,nparts", [(31, 4), (29, 8), (31, 2), (31, 1), (63, 4), (33, 8), (47, 2), (63, 1)])
def test_superkmer_exchange_simulated_shards(mcx, orc, k, nparts):
    """Exchange format v3: every 'rank' turns its reads into per-owner super-k-mer records, every
    owner k-merises the records addressed to it; the union of the (unsharded) owner tables is the
    oracle's graph and every key sits on the shard its canonical minimizer names."""
    import torch
    from mccortex_amd import shard
    assert mcx.superk_supported(k) and not mcx.superk_supported(27) and mcx.superk_supported(63) and not mcx.superk_supported(32)
    rw = mcx.superk_record_words(k)
    W = (2 * k + 63) // 64
    assert rw == 2 * W
    g0 = synth.genome(50000, 77)
    graphs = [mcx.Graph(k, 1, 1 << 20) for _ in range(nparts)]
    all_reads = []
    for r in range(nparts):
        bases, offs = synth.reads(2500, 140, seed=90 + r, g=g0, n_frac=0.08, lower_frac=0.05)
        if r == 0:  # low-complexity reads, short reads, reads of exactly k and k + 1 bases
            hb, ho = orc.pack_reads(["A" * 150] * 50 + ["ACGTTGCA" * 20] * 40 + ["ACG" * 5, "T" * k, "G" * (k + 1), "ACGT" * 9][:4])
            bases = np.concatenate([bases, hb]); offs = np.concatenate([offs, offs[-1] + ho[1:]])
        all_reads.append((bases, offs))
        stream = torch.from_numpy(synth.to_stream(bases, offs)).cuda()
        segs, seg_cap = graphs[r].superk_layout(nparts, stream.numel())
        recs = torch.zeros((nparts, segs, seg_cap, rw), dtype=torch.int64, device="cuda")
        fills = torch.zeros((segs, nparts), dtype=torch.int64, device="cuda")   # replica-major
        graphs[r].superk_bins_dev(stream, stream.numel(), nparts, recs, fills, seg_cap)
        graphs[r].sync()                       # a dropped record would raise here
        counts = fills.t().contiguous()
        assert int(counts.max()) <= seg_cap
        nrec = int(counts.sum())
        assert nrec > 0
        for o in range(nparts):                # "exchange": owner o takes the segments addressed to it
            graphs[o].add_superk_dev(0, recs[o], counts[o], segs, seg_cap, int(counts[o].sum()) * 16)
            graphs[o].sync()
        if r == 1:  # bytes on the wire: well below 8 bytes per occurrence
            st = graphs[r].device_stats()
            assert nrec * 8 * rw < 5.0 * max(1, st.num_kmers_loaded) or nparts == 1
    og = orc.Graph(k, 1, 1 << 21)
    ost = orc.Stats()
    for b, o in all_reads:
        og.add_reads(0, b, o, stats=ost)
    # the owners' counters add up to the whole input's (what the multi-GPU tool all-reduces for the header)
    tot = [g.device_stats() for g in graphs]
    for f in ("num_kmers_loaded", "contigs_parsed", "total_bases_loaded", "num_kmers_novel"):
        assert sum(getattr(t, f) for t in tot) == getattr(ost, f), f
    bodies = []
    for p, g in enumerate(graphs):
        kk, cc, ee = g.records(True)
        for row in kk[::29]:
            assert mcx.superk_owner([int(x) for x in row], k, nparts) == p
        bodies.append(g.export(True))
        g.close()
    assert shard.merge_sorted_bodies(bodies, 8 * W + 5, 8 * W) == og.body_bytes(True)
    if nparts > 1:
        sizes = [len(b) for b in bodies]
        assert min(sizes) > 0.5 * max(sizes)   # minimizer ownership is reasonably balanced


def _distinct_kmer_reads(n_kmers, k, seed):
    """One long random read: ~n_kmers distinct k-mers (plus their edges)."""
    rng = np.random.default_rng(seed)
    n = n_kmers + k - 1
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    # cut into reads of 1000 bases that overlap by k - 1: the same k-mer set, more contigs
    step = 1000 - (k - 1)
    starts = np.arange(0, n - k + 1, step)
    lens = np.minimum(1000, n - starts)
    offs = np.zeros(len(starts) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    idx = np.repeat(starts - offs[:-1].astype(np.int64), lens) + np.arange(int(offs[-1]))
    return np.ascontiguousarray(bases[idx]), offs


@pytest.mark.parametrize("k,defer", [(31, 1), (31, 0), (63, 1)])
def test_high_load_factor_spills_to_overflow_area(mcx, orc, k, defer):
    """hash_table.c:250-281 keeps inserting far beyond 90 % load; here a sub-table that fills up
    spills into the overflow area instead of ending the build.  95 % of a 2^20-slot table."""
    cap = 1 << 20
    bases, offs = _distinct_kmer_reads(int(0.95 * cap), k, seed=5)
    og = orc.Graph(k, 1, 4 * cap)
    og.add_reads(0, bases, offs)
    g = mcx.Graph(k, 1, cap)
    slots, _ = g.capacity()
    assert slots <= cap + cap // 32 + 4096
    g.configure("defer", defer)
    g.add_reads(0, bases, offs)
    g.sync()
    assert g.nkmers == og.nkmers > 0.94 * cap
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    # a second pass finds every key again (sub-table or overflow area): coverage doubles, no new node
    g.add_reads(0, bases, offs)
    g.sync()
    assert g.nkmers == og.nkmers
    og.add_reads(0, bases, offs)
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.close()


def test_really_full_table_is_still_reported(mcx):
    bases, offs = _distinct_kmer_reads(int(1.2 * (1 << 16)), 31, seed=6)
    g = mcx.Graph(31, 1, 1 << 16)
    g.add_reads(0, bases, offs)
    with pytest.raises(mcx.McxError) as ei:
        g.sync()
    assert ei.value.code == mcx.MCX_ERR_FULL and "Hash table is full" in str(ei.value)
    g.close()


@pytest.mark.parametrize("k", [31, 63, 15])
def test_export_in_key_ranges_when_scratch_is_short(mcx, orc, k, monkeypatch):
    """The sorted dump needs ~56 B of scratch per k-mer; when HBM is short it walks the key space in
    ranges of the top bits and splits a range that is fuller than expected (an AT-rich genome puts
    most k-mers under a few prefixes).  MCX_EXPORT_SCRATCH caps what the export may take."""
    rng = np.random.default_rng(k)
    n = 400_000
    g0 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, n, p=[0.45, 0.05, 0.05, 0.45])]  # AT-rich
    bases, offs = synth.reads(6000, 150, seed=k, g=g0, n_frac=0.02)
    og = orc.Graph(k, 1, 1 << 21)
    og.add_reads(0, bases, offs)
    want = og.ctx_bytes(True)[og.header_size():]
    g = mcx.Graph(k, 1, 1 << 21)
    g.add_reads(0, bases, offs)
    full = g.export(True)
    assert full == want
    rs = 8 * g.W + 5
    nk = len(want) // rs
    # room for about 1/8 of the k-mers at a time (+ the fixed part: 2 x 64 MiB record buffers and 64 MiB slack)
    monkeypatch.setenv("MCX_EXPORT_SCRATCH", str(3 * (64 << 20) + nk * 72 // 8))
    assert g.export(True) == want
    a = np.frombuffer(g.export(False), np.uint8).reshape(-1, rs)
    b = np.frombuffer(want, np.uint8).reshape(-1, rs)
    assert sorted(map(bytes, a)) == sorted(map(bytes, b))
    monkeypatch.setenv("MCX_EXPORT_SCRATCH", str(1 << 20))
    with pytest.raises(mcx.McxError):
        g.export(True)
    g.close()


@pytest.mark.parametrize("k,defer", [(31, 1), (63, 1), (21, 0)])
def test_packed_device_stream(mcx, orc, k, defer):
    """mcx_pack_stream_dev + mcx_graph_add_packed_dev: the 3-bit-per-position form of a device stream
    (what the host entry stages) gives the graph of the ASCII stream; odd lengths, N, lower case."""
    import torch
    bases, offs = synth.reads(5000, 133, genome_len=40000, seed=k, n_frac=0.05, lower_frac=0.2, var_len=True)
    stream = torch.from_numpy(synth.to_stream(bases, offs)).cuda()
    n = stream.numel()
    nch = (n + 15) // 16
    code = torch.empty(nch, dtype=torch.int32, device="cuda")
    inv = torch.empty(nch, dtype=torch.int16, device="cuda")
    mcx.pack_stream_dev(stream, n, code, inv)
    torch.cuda.synchronize()
    og = orc.Graph(k, 1, 1 << 20)
    og.add_reads(0, bases, offs)
    g = mcx.Graph(k, 1, 1 << 20)
    g.configure("defer", defer)
    g.add_packed_dev(0, code, inv, n)
    g.sync()
    st = g.device_stats()
    assert g.nkmers == og.nkmers
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.close()


def _structured_kmer_reads(k, seed):
    """k-mers a weak hash places badly: families that differ only in their last 9 bases, only in
    their first 9 bases, and tandem repeats of short periods -- one k-base read per k-mer."""
    rng = np.random.default_rng(seed)
    fams = []
    idx = np.arange(1 << 18, dtype=np.uint64)
    var = np.stack([(idx >> np.uint64(2 * i)) & np.uint64(3) for i in range(9)], axis=1).astype(np.uint8)  # all 4^9 tails
    for fam in range(2):
        fixed = rng.integers(0, 4, k - 9, dtype=np.uint8)
        rows = np.empty((len(idx), k), np.uint8)
        if fam == 0:
            rows[:, :k - 9] = fixed; rows[:, k - 9:] = var   # common prefix, every suffix
        else:
            rows[:, 9:] = fixed; rows[:, :9] = var           # common suffix, every prefix
        fams.append(rows)
    per = []
    for period in range(1, 13):                              # tandem repeats, a few mutations each
        for _ in range(2000):
            unit = rng.integers(0, 4, period, dtype=np.uint8)
            row = np.tile(unit, k // period + 1)[:k].copy()
            pos = rng.integers(0, k, 2)
            row[pos] = rng.integers(0, 4, 2)
            per.append(row)
    fams.append(np.array(per, np.uint8))
    rows = np.concatenate(fams)
    bases = np.frombuffer(b"ACGT", np.uint8)[rows].reshape(-1)
    offs = np.arange(len(rows) + 1, dtype=np.uint64) * k
    return bases, offs


@pytest.mark.parametrize("k", [31, 63])
def test_structured_keys_spread_over_the_sub_tables(mcx, orc, k):
    """The table's own hashes (region / bucket: one multiplication of the folded quotient; sub-table:
    sub_hash) must spread keys that differ in few bits: 0.55 M such k-mers into 2^20 slots (256
    sub-tables of 4096).  A sub-table 1.5 x over its share would spill, the overflow area holds 3 %
    of the table: a skewed hash ends in "Hash table is full"."""
    bases, offs = _structured_kmer_reads(k, seed=9)
    og = orc.Graph(k, 1, 1 << 22)
    og.add_reads(0, bases, offs)
    assert og.nkmers > 500_000
    # 0.62 M slots: load ~0.85 of the hash-addressed slots
    cap = 5 << 17
    for defer in (1, 0):
        g = mcx.Graph(k, 1, cap)
        g.configure("defer", defer)
        g.add_reads(0, bases, offs)
        g.sync()
        assert g.nkmers == og.nkmers
        assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
        g.close()


def test_untouched_sub_tables_are_not_read_only_while_that_is_safe(mcx, orc):
    """TableView::touch: the LDS insert skips reading sub-tables nothing has written since the table
    was zeroed.  Flush, then a direct insert (sets the flag), then another flush: every key of every
    phase must survive; likewise after a reset."""
    k = 31
    parts = [synth.reads(4000, 120, genome_len=150000, seed=s) for s in (21, 22, 23)]
    og = orc.Graph(k, 1, 1 << 22)
    for b, o in parts:
        og.add_reads(0, b, o)
    g = mcx.Graph(k, 1, 1 << 20)
    for rnd in range(2):
        g.configure("defer", 1)
        g.add_reads(0, *parts[0]); g.sync()          # first flush: nothing is read
        g.configure("defer", 0)
        g.add_reads(0, *parts[1]); g.sync()          # direct inserts into sub-tables the flush never touched
        g.configure("defer", 1)
        g.add_reads(0, *parts[2]); g.sync()          # this flush must read them
        assert g.nkmers == og.nkmers
        assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
        g.reset()
    g.close()


@pytest.mark.parametrize("k,ncols,sets", [(31, 4, None), (31, 4, "1"), (31, 4, "3"), (63, 3, None), (21, 6, "4")])
def test_interleaved_colours_keep_their_tuples_apart(mcx, orc, k, ncols, sets, monkeypatch):
    """A population build alternates samples (ctx_build.c:384-407 loads the files of a batch side by
    side; coverage and edges are per colour, db_node.h:240-241).  The partition path keeps a pool of
    L1 bin sets: a colour switch takes another set instead of flushing, and a flush makes one table
    pass per colour over all of that colour's sets (TupleIn::set_map).  Colours 0,1,..,0,1,.. in
    small batches, pool sizes from 1 (a flush per switch, the old behaviour) to 32, several flushes
    (the pool runs out), a device stream in between -- against the oracle, byte for byte."""
    if sets:
        monkeypatch.setenv("MCX_L1_SETS", sets)
    g0 = synth.genome(40000, 17)
    jobs = []
    for i in range(5 * ncols):
        b, o = synth.reads(1500, 110, seed=300 + i, g=g0, n_frac=0.04, err=0.004)
        jobs.append((i % ncols, b, o))
    hb, ho = orc.pack_reads(["C" * 200] * 300)   # a hot k-mer in the last colour: bin overflow -> direct insert
    jobs.append((ncols - 1, hb, ho))
    og, _ = _oracle(orc, k, ncols, jobs)
    want = og.ctx_bytes(True)[og.header_size():]
    for tcap in (1 << 21, 1 << 24):
        g = mcx.Graph(k, ncols, 1 << 20)
        g.configure("defer_tuples", tcap)
        g.configure("profile", 1)
        for col, b, o in jobs:
            g.add_reads(col, b, o)
        assert g.nkmers == og.nkmers, (sets, tcap)
        assert g.export(True) == want, (sets, tcap)
        prof = g.profile()
        assert "k_lds_insert" in prof  # the partition path ran (not the direct fallback)
        if sets is None and tcap == 1 << 24:
            # everything fitted the pool: ONE flush, one table pass (= one k_lds_insert launch per region group) per colour
            groups = prof["k_lds_insert"][0]
            assert groups % ncols == 0 and prof["k_tuples_bin"][0] == groups
        g.close()


def test_device_ceiling_microbenchmarks_and_flush_overlap_switch(mcx, orc):
    """mcx_ubench_stream / mcx_ubench_random_rmw (the ceilings bench.py quotes are measured in its own
    run) return plausible figures for an MI355X, and mcx_graph_configure("flush_overlap", 0 | 1) -- the switch behind
    bench.py's isolated per-kernel pass -- does not change the graph."""
    st = mcx.ubench_stream(1 << 30)
    assert 1000 < st["copy"] < 8000 and 1000 < st["read"] < 8000 and 1000 < st["write"] < 8000   # GB/s; 8 TB/s is the datasheet
    rw = mcx.ubench_random_rmw(1 << 30, 1 << 24)
    assert 1e9 < rw["rmw"] < 1e11 and rw["load16"] > rw["rmw"]
    with pytest.raises(mcx.McxError):
        mcx.ubench_stream(1 << 20)
    g0 = synth.genome(300000, 3)
    b, o = synth.reads(60000, 120, seed=11, g=g0, n_frac=0.02)
    og = orc.Graph(31, 1, 1 << 22)
    og.add_reads(0, b, o)
    want = og.body_bytes(True)
    for ov in (0, 1):
        g = mcx.Graph(31, 1, 1 << 24)      # 8 regions: two flush groups with flush_regions = 4 (the overlap needs two)
        g.configure("flush_regions", 4)
        g.configure("flush_overlap", ov)
        g.add_reads(0, b, o)
        assert g.export(True) == want
        g.close()


def test_batch_without_kmers_then_another_colour(mcx, orc, monkeypatch):
    """A batch that yields no k-mer at all (reads shorter than k) leaves its bin set bound to its colour with nothing
    on the books once its launch has settled; with ONE bin set the next colour needs that set back, and the flush that
    would release it has nothing to flush.  Seed 210 of the round-4 soak: "no L1 bin set after a flush"."""
    import time
    monkeypatch.setenv("MCX_L1_SETS", "1")
    short_b, short_o = synth.reads(66, 17, genome_len=3000, seed=5)
    big_b, big_o = synth.reads(1600, 170, genome_len=40_000, seed=6, n_frac=0.02)
    og = orc.Graph(29, 2, 1 << 20)
    og.add_reads(1, short_b, short_o)
    og.add_reads(0, big_b, big_o)
    g = mcx.Graph(29, 2, 1 << 20)
    g.configure("defer_tuples", 1 << 20)
    g.add_reads(1, short_b, short_o)
    time.sleep(0.3)              # (the launch settles: its counter snapshot reaches the host)
    g.add_reads(0, big_b, big_o)
    g.sync()
    assert g.export(True) == og.ctx_bytes(True)[og.header_size():]
    g.close()


@pytest.mark.parametrize("k", [31, 63])
@pytest.mark.parametrize("how", ["deferred", "direct", "2 shards v3", "2 shards v2"])
def test_coverage_saturates_through_the_read_path(mcx, orc, k, how, monkeypatch):
    """db_node_increment_coverage (src/graph/db_node.c:139-144) stops at UINT32_MAX.  A node is loaded from a
    record with coverage 0xFFFFFFFE (0xFFFFFFFD, 0xFFFFFFFF) and reads then hit it three times -- through the
    partition + LDS insert, the direct HBM atomics and two shards in either exchange format: the exported
    coverage is 0xFFFFFFFF, never a wrapped count, and everything else counts on."""
    import struct
    read = "ACGGTCATTGACCGATTACGGCATCGGATTCAGCTTAGGCATCGATTGCAGCTAGCTAGGCTTAACGGCTAGCATCGGATCATTC"
    W = (2 * k + 63) // 64
    nk = len(read) - k + 1
    keys = [mcx.kmer_canonical(mcx.kmer_from_str(read[i:i + k], k), k)[0] for i in range(nk)]
    near = [0xFFFFFFFE, 0xFFFFFFFD, 0xFFFFFFFF, 0xFFFFFFFC, 5]
    rec = b"".join(struct.pack("<%dQIB" % W, *(key + [near[i % len(near)], 0x10 if i % 2 else 0x00])) for i, key in enumerate(keys))
    other, ooffs = synth.reads(3000, 100, genome_len=20000, seed=k)
    rb, ro = orc.pack_reads([read] * 3 + [read[5:]])
    og = orc.Graph(k, 1, 1 << 20)
    if how.startswith("2 shards"):
        monkeypatch.setenv("MCX_MULTI_EXCHANGE", how[-2:])
        g = mcx.Graph(k, 1, 1 << 20, devices=[0, 0])
    else:
        g = mcx.Graph(k, 1, 1 << 20)
        if how == "direct":
            g.configure("defer", 0)
    og.add_reads(0, other, ooffs); g.add_reads(0, other, ooffs)
    for i, key in enumerate(keys):
        og.add_record(key, [near[i % len(near)]], [0x10 if i % 2 else 0x00])
    g.add_records(rec, 1, [(0, 0)])
    og.add_reads(0, rb, ro); g.add_reads(0, rb, ro)
    body = g.export(True)
    assert body == og.body_bytes(True)
    rs = 8 * W + 5
    recs = {bytes(body[i:i + 8 * W]): struct.unpack_from("<I", body, i + 8 * W)[0] for i in range(0, len(body), rs)}
    sat = [recs[struct.pack("<%dQ" % W, *key)] for i, key in enumerate(keys) if near[i % len(near)] >= 0xFFFFFFFC and i >= 5]
    assert sat and all(c == 0xFFFFFFFF for c in sat)   # (the first 5 k-mers got 3 hits, the rest 4)
    g.close()


@pytest.mark.parametrize("defer", [1, 0])
def test_overfull_table_fails_fast(mcx, defer):
    """-n far too small for the input: the reference dies at the first insert that finds no slot ("Hash table is
    full", hash_table.c:119-123).  Here every further new key used to scan the whole overflow area (1 / 32 of the
    table) before giving up -- a build of 4.8 G distinct k-mers into 1.1 G slots did not finish in a quarter of an
    hour.  The first failed scan now raises a flag that turns the later ones away: MCX_ERR_FULL within seconds."""
    import time
    rng = np.random.default_rng(4)
    n, L = 300_000, 150
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)]       # iid: 36 M distinct k-mers
    offs = np.arange(n + 1, dtype=np.uint64) * L
    g = mcx.Graph(31, 1, 1 << 22)                                            # 4 M slots (+ 1 / 32 overflow area)
    g.configure("defer", defer)
    t0 = time.perf_counter()
    with pytest.raises(mcx.McxError) as ei:
        g.add_reads(0, bases, offs)
        g.sync()
    assert ei.value.code == mcx.MCX_ERR_FULL
    assert time.perf_counter() - t0 < 30
    g.close()
