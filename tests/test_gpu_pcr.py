"""build --remove-pcr on the GPU (mcx_graph_add_reads_pcr) against the restatement walking the reads in order."""
import os

import numpy as np
import pytest

import pcr_cases as pc
import synth

pytestmark = pytest.mark.gpu

FIELDS = ("num_se_reads", "num_good_reads", "num_bad_reads", "total_bases_read", "total_bases_loaded",
          "contigs_parsed", "num_kmers_loaded", "num_kmers_novel")


def _dev_stats(g, st):
    cur = g.device_stats()
    for f in ("num_good_reads", "num_bad_reads", "total_bases_loaded", "contigs_parsed", "num_kmers_loaded", "num_kmers_novel"):
        setattr(st, f, getattr(cur, f))
    return st


def _same_graph(mcx, g, og, k, ncols):
    assert g.nkmers == og.nkmers
    want = og.ctx_bytes(True)
    assert g.export(True) == want[og.header_size():]


def test_reference_pcr_duplicate_unit_test(mcx, orc):
    """src/tests/build_graph_tests.c:19-148 call by call; every call is its own batch"""
    g = mcx.Graph(pc.K, 1, 1024)
    og = orc.Graph(pc.K, 1, 1024)
    st, ot = mcx.LoadStats(), orc.Stats()
    odup = [0, 0, 0]
    for r1, r2, md, pcr, (c1, c2) in pc.STEPS:
        b, o = orc.pack_reads([r1] if r2 is None else [r1, r2])
        if pcr:
            g.add_reads_pcr(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF, paired=r2 is not None, matedir=md, stats=st)
            _, d = og.add_reads_pcr(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF, paired=r2 is not None, matedir=md, stats=ot)
            odup = [x + y for x, y in zip(odup, d)]
        else:
            g.add_reads(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF, stats=st)
            og.add_reads(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF, stats=ot)
        for kmer, c in ((pc.K1, c1), (pc.K2, c2)):   # what the reference's test asserts after this call ...
            if c is not None:
                assert og.lookup(kmer)[0][0] == c, (r1, r2, md)
        _same_graph(mcx, g, og, pc.K, 1)             # ... holds for the GPU graph: it is the same graph
    _dev_stats(g, st)
    assert (st.num_dup_se_reads, st.num_dup_pe_pairs, st.num_pe_reads) == tuple(odup) == (2, 6, 16)
    assert {f: getattr(st, f) for f in FIELDS} == {f: getattr(ot, f) for f in FIELDS}
    assert st.total_bases_loaded == pc.TOTAL_SEQ and st.contigs_parsed == pc.CONTIGS
    _same_graph(mcx, g, og, pc.K, 1)
    g.close()


def _dupy_reads(n, readlen, seed, genome_len=600, **kw):
    """reads from a tiny genome: many share a start (and strand), i.e. are duplicates of each other"""
    return synth.reads(n, readlen, genome_len=genome_len, seed=seed, **kw)


@pytest.mark.parametrize("k,paired,matedir,nbatches", [
    (31, False, "FF", 1), (31, False, "RR", 3), (31, True, "FR", 1), (31, True, "RF", 4),
    (63, True, "FF", 2), (63, False, "FR", 1), (21, True, "RR", 1), (3, True, "FR", 2)])
def test_filter_matches_walk_in_input_order(mcx, orc, k, paired, matedir, nbatches):
    n = 6000
    bases, offs = _dupy_reads(n, 90, seed=k + nbatches, n_frac=0.1, lower_frac=0.1, var_len=(nbatches > 1))
    g = mcx.Graph(k, 1, 1 << 16)
    og = orc.Graph(k, 1, 1 << 16)
    st, ot = mcx.LoadStats(), orc.Stats()
    odup = [0, 0, 0]
    cuts = [2 * (n // 2 * i // nbatches) for i in range(nbatches + 1)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = offs[lo:hi + 1]
        g.add_reads_pcr(0, bases, o, paired=paired, matedir=matedir, stats=st)
        _, d = og.add_reads_pcr(0, bases, o, paired=paired, matedir=matedir, stats=ot)
        odup = [x + y for x, y in zip(odup, d)]
    _dev_stats(g, st)
    assert (st.num_dup_se_reads, st.num_dup_pe_pairs, st.num_pe_reads) == tuple(odup)
    assert odup[1 if paired else 0] > n // 20          # the input does hold duplicates ...
    assert st.num_good_reads > (n // 20 if k > 5 else 10)  # ... and reads that are not (k = 3: 128 starts in all)
    assert {f: getattr(st, f) for f in FIELDS} == {f: getattr(ot, f) for f in FIELDS}
    _same_graph(mcx, g, og, k, 1)
    g.close()


def test_filter_with_quality_and_homopolymer_cutoffs(mcx, orc):
    """the start k-mer is the one seq_contig_start finds under -Q / -H; mates carry their own cutoff"""
    n, k = 4000, 31
    bases, offs = _dupy_reads(n, 100, seed=5, n_frac=0.05)
    rng = np.random.default_rng(9)
    quals = rng.integers(33, 74, len(bases)).astype(np.uint8)
    quals[rng.random(len(bases)) < 0.9] = 70
    for fq1, fq2, hp in ((60, 60, 0), (55, 64, 4), (0, 0, 3)):
        g = mcx.Graph(k, 1, 1 << 16)
        og = orc.Graph(k, 1, 1 << 16)
        st, ot = mcx.LoadStats(), orc.Stats()
        g.add_reads_pcr(0, bases, offs, quals=quals, fq_cutoff=fq1, fq_cutoff2=fq2, hp_cutoff=hp, paired=True, matedir="FR", stats=st)
        _, d = og.add_reads_pcr(0, bases, offs, quals=quals, fq_cutoff=fq1, fq_cutoff2=fq2, hp_cutoff=hp, paired=True, matedir="FR", stats=ot)
        _dev_stats(g, st)
        assert (st.num_dup_se_reads, st.num_dup_pe_pairs, st.num_pe_reads) == tuple(d) and d[1] > 0
        assert {f: getattr(st, f) for f in FIELDS} == {f: getattr(ot, f) for f in FIELDS}
        _same_graph(mcx, g, og, k, 1)
        g.close()


def test_reset_between_colours_and_plain_reads_in_between(mcx, orc):
    """readstrt is wiped when the colour changes (ctx_build.c:389-395); unfiltered loads do not touch it"""
    k = 31
    bases, offs = _dupy_reads(3000, 80, seed=11)
    g = mcx.Graph(k, 2, 1 << 16)
    og = orc.Graph(k, 2, 1 << 16)
    for col, reset in ((0, False), (0, False), (1, True), (1, False)):
        if reset:
            g.pcr_reset(); og.pcr_reset()
        a = g.add_reads_pcr(col, bases, offs, matedir="FF")
        _, d = og.add_reads_pcr(col, bases, offs, matedir="FF")
        assert a.num_dup_se_reads == d[0]
        g.add_reads(col, bases[:int(offs[100])], offs[:101])
        og.add_reads(col, bases[:int(offs[100])], offs[:101])
    _same_graph(mcx, g, og, k, 2)
    # mcx_graph_reset forgets the read starts together with the graph
    g.reset()
    og2 = orc.Graph(k, 2, 1 << 16)
    a = g.add_reads_pcr(0, bases, offs, matedir="FF")
    _, d = og2.add_reads_pcr(0, bases, offs, matedir="FF")
    assert a.num_dup_se_reads == d[0]
    _same_graph(mcx, g, og2, k, 2)
    g.close()


def test_argument_errors(mcx, orc):
    g = mcx.Graph(31, 1, 1024)
    b, o = orc.pack_reads(["ACGT" * 10] * 3)
    with pytest.raises(mcx.McxError):
        g.add_reads_pcr(0, b, o, paired=True)          # mates come in twos
    with pytest.raises(mcx.McxError):
        g.add_reads_pcr(0, b, o, matedir=7)
    with pytest.raises(mcx.McxError):
        g.add_reads_pcr(1, b, o)
    g.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MCX_FUZZ_SEEDS_BYTES", "6"))))
def test_fuzz_arbitrary_bytes_lengths_and_cutoffs(mcx, orc, seed):
    """reads of arbitrary bytes and lengths (0 .. 3 k), arbitrary qualities, random preferences, in several batches"""
    rng = np.random.default_rng(300 + seed)
    k = int(rng.choice([5, 21, 31, 33, 63]))
    alphabet = np.concatenate([np.frombuffer(b"ACGT" * 30 + b"acgtNn-*", np.uint8), rng.integers(0, 256, 8).astype(np.uint8)])
    n = 2 * int(rng.integers(300, 1500))
    starts = [bytes(rng.choice(alphabet[:120], k + 3)) for _ in range(40)]     # few distinct read starts -> duplicates
    reads = []
    for i in range(n):
        body = bytes(rng.choice(alphabet, int(rng.integers(0, 3 * k))))
        r = starts[int(rng.integers(0, len(starts)))] + body if rng.random() < 0.8 else body
        if rng.random() < 0.3:      # some as reverse complements: the other strand is another start
            r = r[::-1].translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))
        reads.append(r)
    bases, offs = orc.pack_reads(reads)
    quals = rng.integers(30, 80, len(bases)).astype(np.uint8)
    paired = bool(rng.integers(0, 2))
    matedir = ["FF", "FR", "RF", "RR"][int(rng.integers(0, 4))]
    fq1, fq2, hp = (int(rng.choice([0, 35, 50])), int(rng.choice([0, 40, 60])), int(rng.choice([0, 0, 3, 5])))
    g = mcx.Graph(k, 1, 1 << 16)
    og = orc.Graph(k, 1, 1 << 16)
    st, ot = mcx.LoadStats(), orc.Stats()
    odup = [0, 0, 0]
    nb = int(rng.integers(1, 5))
    cuts = [2 * (n // 2 * i // nb) for i in range(nb + 1)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        kw = dict(quals=quals, fq_cutoff=fq1, fq_cutoff2=fq2, hp_cutoff=hp, paired=paired, matedir=matedir)
        g.add_reads_pcr(0, bases, offs[lo:hi + 1], stats=st, **kw)
        _, d = og.add_reads_pcr(0, bases, offs[lo:hi + 1], stats=ot, **kw)
        odup = [x + y for x, y in zip(odup, d)]
    _dev_stats(g, st)
    assert (st.num_dup_se_reads, st.num_dup_pe_pairs, st.num_pe_reads) == tuple(odup), (k, paired, matedir, fq1, fq2, hp)
    assert {f: getattr(st, f) for f in FIELDS} == {f: getattr(ot, f) for f in FIELDS}
    _same_graph(mcx, g, og, k, 1)
    g.close()
