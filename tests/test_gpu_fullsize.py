"""Parity at BASELINE.json's full sizes.

Byte-wise comparison of multi-GB exports is impractical, so a graph is reduced to an order-independent
checksum of its exported records (mcx_graph_checksum; pinned to the oracle's records at small size below).

* Against the ORACLE at the benchmark's own table geometry (2^30 slots = 512 regions x 512 sub-tables, the
  shape every bench.py number is measured on): one C2 step (5 M reads, 600 M occurrences; the oracle takes
  about 20 s with 32 threads) for k = 31 and k = 63, through the partition + LDS-insert path, the direct
  HBM-atomics path and eight in-process shards -- checksum, node count and loading statistics must equal
  those of the oracle's records.  (bench.py's cpu_baseline leg does the same on its timed oracle run, up to
  the whole 50 M-read set with --oracle-steps 10: `config.checksum_matches_oracle`.)
* At the whole 6 G occurrences, where the oracle needs minutes rather than seconds, independent ways of
  building the same graph must agree on the checksum: the partition + LDS-insert path with one flush and
  with many, the direct HBM-atomics path, and the sharded builds (exchange formats v2 and v3, shards
  simulated on one GPU: the sum of the shard checksums is the checksum of the union).  Plus counting
  identities: every occurrence is counted exactly once (sum of coverage == k-mers loaded), every node
  has coverage (histogram bin 0 empty), node count == novel counter."""
import os
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M64 = (1 << 64) - 1


@pytest.mark.parametrize("k,ncols", [(31, 1), (63, 1), (31, 3), (45, 2)])
def test_checksum_is_pinned_to_oracle_records(mcx, orc, k, ncols):
    g0 = synth.genome(20000, 5)
    og = orc.Graph(k, ncols, 1 << 20)
    g = mcx.Graph(k, ncols, 1 << 20)
    for c in range(ncols):
        b, o = synth.reads(2500, 110, seed=70 + c, g=g0, n_frac=0.05)
        og.add_reads(c, b, o); g.add_reads(c, b, o)
    body = og.body_bytes(True)
    want = mcx.records_checksum(body, k, ncols)
    cs, n = g.checksum()
    assert (cs, n) == (want, og.nkmers)
    assert mcx.records_checksum(g.export(False), k, ncols) == want      # order does not matter
    # the checksum sees every field: flip one edge bit / one coverage
    rs = 8 * og.W + 5 * ncols
    bad = bytearray(body); bad[rs - 1] ^= 1
    assert mcx.records_checksum(bytes(bad), k, ncols) != want
    bad = bytearray(body); bad[8 * og.W] ^= 1
    assert mcx.records_checksum(bytes(bad), k, ncols) != want
    g.close()


@pytest.fixture(scope="module")
def c2_batches():
    """BASELINE config C2: 50 M x 150 bp reads from a 200 Mbp random genome, as bench.py makes them"""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    genome = bench.make_genome(bench.GENOME_PER_GPU, dev, seed=42)
    batches = [bench.make_batch(genome, bench.BATCH_READS, seed=1000 + i, device=dev) for i in range(10)]
    del genome
    torch.cuda.empty_cache()
    yield batches
    del batches
    torch.cuda.empty_cache()


def _build(mcx, batches, k, ncols, slots, cfg, colours=None):
    g = mcx.Graph(k, ncols, slots)
    for key, v in cfg.items():
        g.configure(key, v)
    for i, b in enumerate(batches):
        g.add_stream_dev(colours[i] if colours else 0, b, b.numel())
    g.sync()
    st = g.device_stats()
    cs, n = g.checksum()
    nk, sc = g.kmer_covg()
    hist = g.covg_histogram(64)
    ist = g.insert_stats()
    # no occurrence of these well-spread inputs may leave the fast path (bin overflow -> per-occurrence insert)
    assert ist["fallback_inserts"] == 0 and ist["foreign_inserts"] == 0, ist
    out = dict(checksum=cs, nodes=n, nkmers=g.nkmers, loaded=st.num_kmers_loaded, contigs=st.contigs_parsed,
               covg_nodes=nk.tolist(), covg_sum=sc.tolist(), hist0=int(hist[0]), hist_total=int(hist.sum()))
    g.close()
    return out


@pytest.mark.parametrize("k", [31, 63])
def test_benchmark_geometry_matches_oracle(mcx, orc, c2_batches, k):
    """GPU vs ORACLE at the table geometry of the benchmark (2^30 slots: lb1 = 9, 512 sub-tables per region;
    k = 63: 1024 sub-tables of 2048 slots), on the first C2 step.  The direct and the deferred path share
    their front end and the quotient addressing, so their agreement alone would not exclude a common error
    that only shows at this geometry: the oracle shares nothing with them."""
    import bench
    SLOTS = 1 << 30
    b = c2_batches[0]
    n = bench.BATCH_READS
    host = b.reshape(n, bench.READ_LEN + 1)[:, :bench.READ_LEN].contiguous().cpu().numpy().reshape(-1)
    offs = np.arange(n + 1, dtype=np.uint64) * bench.READ_LEN
    nt = min(32, os.cpu_count() or 1)
    og = orc.Graph(k, 1, 1 << 29)      # (the oracle's own capacity does not enter its records)
    og.tune(nt)
    st = og.add_reads(0, host, offs, nthreads=nt)
    body = og.body_array(False)
    want = (mcx.records_checksum(body, k, 1), og.nkmers)
    assert body.size == og.nkmers * (8 * og.W + 5)
    del body, og
    assert want[1] > 150_000_000
    for name, kw, cfg in (("deferred", {}, {"defer_tuples": 1_000_000_000}),
                          ("deferred, placed bins", {}, {"place_bins": 3, "defer_tuples": 1_000_000_000}),  # (two allocations, one per half of the flush overlap)
                          ("direct", {}, {"defer": 0}),
                          ("8 in-process shards", {"devices": [0] * 8}, {"defer_tuples": 125_000_000})):
        g = mcx.Graph(k, 1, SLOTS, **kw)
        for key, v in cfg.items():
            g.configure(key, v)
        g.add_stream_dev(0, b, b.numel())
        g.sync()
        ds = g.device_stats()
        got = g.checksum()
        nk, sc = g.kmer_covg()
        g.close()
        assert got == want, (k, name)
        assert (ds.num_kmers_loaded, ds.contigs_parsed, ds.num_kmers_novel) == (st.num_kmers_loaded, st.contigs_parsed, st.num_kmers_novel), (k, name)
        assert (int(nk[0]), int(sc[0])) == (want[1], st.num_kmers_loaded), (k, name)


def test_benchmark_geometry_four_colours_matches_oracle(mcx, orc, c2_batches):
    """BASELINE config C5's shape at the benchmark's table geometry, against the ORACLE: one C2 step dealt to four
    colours (sample c = reads [c n / 4, (c + 1) n / 4)), 2^30 slots, through the partition + LDS insert (a pool of bin
    sets, one table pass per colour), the direct path and eight in-process shards (C5 names 8 GPUs)."""
    import bench
    k, ncols, SLOTS = 31, 4, 1 << 30
    b = c2_batches[0]
    n = bench.BATCH_READS
    rows = b.reshape(n, bench.READ_LEN + 1)
    host = rows[:, :bench.READ_LEN].contiguous().cpu().numpy()
    nt = min(32, os.cpu_count() or 1)
    og = orc.Graph(k, ncols, 1 << 29)
    og.tune(nt)
    loaded = 0
    cut = [n * c // ncols for c in range(ncols + 1)]
    for c in range(ncols):
        part = host[cut[c]:cut[c + 1]].reshape(-1)
        offs = np.arange(cut[c + 1] - cut[c] + 1, dtype=np.uint64) * bench.READ_LEN
        loaded += og.add_reads(c, part, offs, nthreads=nt).num_kmers_loaded
    body = og.body_array(False)
    want = (mcx.records_checksum(body, k, ncols), og.nkmers)
    del body, og
    # (pieces stay alive until the graph has consumed them: add_stream_dev is asynchronous on the graph's own stream)
    pieces = []
    for c in range(ncols):
        mid = (cut[c] + cut[c + 1]) // 2
        pieces.append((rows[cut[c]:mid].reshape(-1).clone(), rows[mid:cut[c + 1]].reshape(-1).clone()))
    for name, kw, cfg in (("deferred", {}, {}), ("direct", {}, {"defer": 0}), ("8 in-process shards", {"devices": [0] * 8}, {"defer_tuples": 125_000_000})):   # (all eight share the one device's HBM here)
        g = mcx.Graph(k, ncols, SLOTS, **kw)
        for key, v in cfg.items():
            g.configure(key, v)
        for rep in range(2):            # the samples alternate: 0 1 2 3 0 1 2 3 (halves of every quarter)
            for c in range(ncols):
                g.add_stream_dev(c, pieces[c][rep], pieces[c][rep].numel())
        g.sync()
        ds = g.device_stats()
        got = g.checksum()
        nk, sc = g.kmer_covg()
        ist = g.insert_stats()
        g.close()
        assert got == want, name
        assert ds.num_kmers_loaded == loaded and int(sum(int(x) for x in sc)) == loaded, name
        assert ist["fallback_inserts"] == 0, (name, ist)


def test_huge_table_geometry_matches_oracle(mcx, orc):
    """GPU vs ORACLE on a table beyond 2^32 slots (2^33 slots = 128 GiB: 1024 regions x 2048 sub-tables, the C2-stress
    shape): the split into 2048 sub-table bins runs with 1024-thread blocks and tiles of 16384 tuples (round 4), the
    k-merising kernel with 1024 region bins.  300 k reads; both split geometries must give the oracle's records."""
    import subprocess
    b, o = synth.reads(300_000, 150, genome_len=3_000_000, seed=33, n_frac=0.02)
    og = orc.Graph(31, 1, 1 << 26)
    st = og.add_reads(0, b, o)
    body = og.body_array(False)
    want = (mcx.records_checksum(body, 31, 1), og.nkmers)
    del body, og
    g = mcx.Graph(31, 1, 1 << 33)
    g.add_reads(0, b, o)
    g.sync()
    ds = g.device_stats()
    got = g.checksum()
    prof_ok = g.nkmers == want[1]
    g.close()
    assert got == want and prof_ok
    assert (ds.num_kmers_loaded, ds.contigs_parsed) == (st.num_kmers_loaded, st.contigs_parsed)
    # ... and the 256-thread split of the same bins (MCX_SPLIT_T=256), in a process of its own (the choice is read once)
    prog = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import synth, mccortex_amd as mcx; "
            "b, o = synth.reads(300_000, 150, genome_len=3_000_000, seed=33, n_frac=0.02); g = mcx.Graph(31, 1, 1 << 33); "
            "g.add_reads(0, b, o); g.sync(); print('CS', *g.checksum())") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, MCX_SPLIT_T="256"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-1500:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("CS")][-1].split()
    assert (int(line[1]), int(line[2])) == want


def test_c2_full_size_all_paths_agree(mcx, c2_batches):
    import torch
    from mccortex_amd import shard
    K, SLOTS = 31, 1 << 30
    ref = _build(mcx, c2_batches, K, 1, SLOTS, {"defer_tuples": 8_000_000_000})
    assert ref["loaded"] == 5_987_568_952 or ref["loaded"] > 5.9e9        # ~120 k-mers per read minus the N reads
    assert ref["nodes"] == ref["nkmers"] == ref["covg_nodes"][0] == ref["hist_total"]
    assert ref["covg_sum"][0] == ref["loaded"] and ref["hist0"] == 0
    many = _build(mcx, c2_batches, K, 1, SLOTS, {"defer_tuples": 700_000_000, "flush_regions": 64})   # 9-10 flushes
    direct = _build(mcx, c2_batches, K, 1, SLOTS, {"defer": 0})
    for other in (many, direct):
        assert other == ref
    # sharded, exchange format v3: 4 owners with ordinary tables of a quarter of the size
    N = 4
    owners = [mcx.Graph(K, 1, SLOTS // N) for _ in range(N)]
    for o in owners:
        o.configure("defer_tuples", 2_500_000_000)
    segs, cap = owners[0].superk_layout(N, c2_batches[0].numel())
    recs = torch.empty((N, segs, cap, 2), dtype=torch.int64, device="cuda")
    fills = torch.zeros((segs, N), dtype=torch.int64, device="cuda")
    sender = owners[0]
    loaded0 = 0
    for b in c2_batches:
        fills.zero_()
        sender.superk_bins_dev(b, b.numel(), N, recs, fills, cap)
        sender.sync()                      # the owners run on their own streams
        counts = fills.t().contiguous()
        assert int(fills.max()) <= cap
        for p, o in enumerate(owners):
            o.add_superk_dev(0, recs[p], counts[p], segs, cap, int(counts[p].sum()) * 16)
        torch.cuda.synchronize()
        for o in owners:
            o.sync()
    loaded0 = sender.device_stats().num_kmers_loaded
    parts = [o.checksum() for o in owners]
    sums = [o.kmer_covg()[1][0] for o in owners]
    for o in owners:
        o.close()
    assert loaded0 == ref["loaded"]
    assert (sum(c for c, _ in parts) & M64, sum(n for _, n in parts)) == (ref["checksum"], ref["nodes"])
    assert int(sum(int(x) for x in sums)) == ref["loaded"]
    assert max(n for _, n in parts) < 1.05 * ref["nodes"] / N              # balanced ownership
    # sharded, exchange format v2: 2 shards of the quotient-hash-prefix sharded table
    N = 2
    shards = [mcx.Graph(K, 1, SLOTS // N, nparts=N, part=p) for p in range(N)]
    for o in shards:
        o.configure("defer_tuples", 4_000_000_000)
    ntup = c2_batches[0].numel()
    segs, seg_cap, ov_cap = shards[0].shard_layout(ntup * 120 // 151)
    blk = shard.BlockExchange(N, segs, seg_cap, ov_cap, 1, "cuda")
    for b in c2_batches:
        blk.zero_counts()
        blk.fill(shards[0], b, b.numel())
        shards[0].sync()                   # the other shard runs on its own stream
        for p, o in enumerate(shards):
            rb = shard.BlockExchange(1, segs, seg_cap, ov_cap, 1, "cuda")
            rb.keys, rb.counts = blk.keys[p:p + 1], blk.counts[p:p + 1]
            rb.ov_keys, rb.ov_edges, rb.ov_counts = blk.ov_keys[p:p + 1], blk.ov_edges[p:p + 1], blk.ov_counts[p:p + 1]
            rb.consume(o, 0, ntup)
        torch.cuda.synchronize()
        assert not blk.overflowed()
        for o in shards:
            o.sync()
    parts = [o.checksum() for o in shards]
    for o in shards:
        o.close()
    assert (sum(c for c, _ in parts) & M64, sum(n for _, n in parts)) == (ref["checksum"], ref["nodes"])


def test_c4_and_c5_full_size_paths_agree(mcx, c2_batches):
    # C4: k = 63 on the same reads (4.4 G occurrences)
    a = _build(mcx, c2_batches, 63, 1, 1 << 30, {"defer_tuples": 5_000_000_000})
    b = _build(mcx, c2_batches, 63, 1, 1 << 30, {"defer": 0})
    assert a == b and a["covg_sum"][0] == a["loaded"] and a["hist0"] == 0 and a["loaded"] > 4.3e9
    # C4 sharded, exchange format v3 with the 32-byte records of two-word keys: 4 owners, checksum == fused
    import torch
    N, K = 4, 63
    owners = [mcx.Graph(K, 1, (1 << 30) // N) for _ in range(N)]
    for o in owners:
        o.configure("defer_tuples", 2_000_000_000)
    segs, cap = owners[0].superk_layout(N, c2_batches[0].numel())
    recs = torch.empty((N, segs, cap, mcx.superk_record_words(K)), dtype=torch.int64, device="cuda")
    fills = torch.zeros((segs, N), dtype=torch.int64, device="cuda")
    nrec = 0
    for bt in c2_batches:
        fills.zero_()
        owners[0].superk_bins_dev(bt, bt.numel(), N, recs, fills, cap)
        owners[0].sync()
        counts = fills.t().contiguous()
        assert int(fills.max()) <= cap
        nrec += int(counts.sum())
        for p, o in enumerate(owners):
            o.add_superk_dev(0, recs[p], counts[p], segs, cap, int(counts[p].sum()) * 16)
        torch.cuda.synchronize()
        for o in owners:
            o.sync()
    assert owners[0].device_stats().num_kmers_loaded == a["loaded"]
    parts = [o.checksum() for o in owners]
    for o in owners:
        o.close()
    del recs
    torch.cuda.empty_cache()
    assert (sum(c for c, _ in parts) & M64, sum(n for _, n in parts)) == (a["checksum"], a["nodes"])
    assert nrec * 32 < 5.0 * a["loaded"]  # bytes on the links: below 5 per occurrence (format v2: 17)
    # C5-like: the 10 batches spread over 4 colours
    cols = [0, 0, 0, 1, 1, 1, 2, 2, 3, 3]
    a = _build(mcx, c2_batches, 31, 4, 1 << 30, {}, cols)
    b = _build(mcx, c2_batches, 31, 4, 1 << 30, {"defer": 0}, cols)
    assert a == b and sum(a["covg_sum"]) == a["loaded"] and a["hist0"] == 0
    assert all(n > 0 for n in a["covg_nodes"])


def test_c2_full_size_in_process_multi(mcx, c2_batches):
    """mcx_graph_create_multi with two, four and eight (C3's count) shards on the one GPU at C2 size: the
    facade's checksum / node count / counters equal the single table's (csrc/mcx_multi.h)."""
    ref = _build(mcx, c2_batches, 31, 1, 1 << 30, {"defer_tuples": 8_000_000_000})
    # exchange format v3 (minimizer-owned super-k-mer records: the default at k = 31) and v2 (hash-prefix shards)
    for xch, devs in (("v3", [0, 0]), ("v3", [0, 0, 0, 0]), ("v3", [0] * 8), ("v2", [0, 0]), ("v2", [0] * 8)):
        os.environ["MCX_MULTI_EXCHANGE"] = xch
        g = mcx.Graph(31, 1, 1 << 30, devices=devs)
        os.environ.pop("MCX_MULTI_EXCHANGE")
        g.configure("defer_tuples", 8_000_000_000 // len(devs))
        for b in c2_batches:
            g.add_stream_dev(0, b, b.numel())
        g.sync()
        st = g.device_stats()
        cs, n = g.checksum()
        nk, sc = g.kmer_covg()
        ist = g.insert_stats()
        g.close()
        assert ist["fallback_inserts"] == 0 and ist["foreign_inserts"] == 0 and ist["spilled"] == 0, (xch, devs, ist)
        assert (cs, n, st.num_kmers_loaded, st.contigs_parsed) == (ref["checksum"], ref["nodes"], ref["loaded"], ref["contigs"]), (xch, devs)
        assert int(nk[0]) == ref["nodes"] and int(sc[0]) == ref["loaded"]
