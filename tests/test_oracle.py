"""CPU tests of the oracle: pins it against the reference's recorded known answers, the
reference's vendored lookup3 (oracle/_ref), and the invariants of the reference's unit tests."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import synth

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _words(x, W):
    return [int(x.b[i]) for i in range(W)]


def _rand_kmer(rng, k):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, k))


@pytest.mark.parametrize("name,k", [("kmer_k31", 31), ("kmer_k63", 63)])
def test_kmer_kats(orc, name, k):
    L = orc.lib()
    kat = KATS[name]
    W = L.orc_words_for_k(k)
    x = L.orc_kmer_from_str(kat["seq"].encode(), k)
    assert _words(x, W) == [int(w, 16) for w in kat["words"]]
    buf = C.create_string_buffer(128)
    L.orc_kmer_to_str(L.orc_kmer_revcomp(x, k), k, buf)
    assert buf.value.decode() == kat["revcomp"]
    key = L.orc_kmer_get_key(x, k)
    for iv, h in kat["hash"].items():
        assert L.orc_kmer_hash(key, k, int(iv)) == int(h, 16)
    if kat.get("canonical_is_forward"):
        assert _words(key, W) == _words(x, W)


def test_hash_matches_reference_lookup3(orc):
    """kmer_hash.h is a fixed-length specialisation of libs/misc/lookup3.h hashlittle: compare
    the oracle with the reference's own vendored file compiled into oracle/_ref."""
    R = orc.ref_lookup3()
    if R is None:
        pytest.skip("oracle/_ref/liblk3ref.so not built (needs /root/reference at build time)")
    L = orc.lib()
    rng = np.random.default_rng(1)
    for k in (3, 15, 31, 33, 47, 63, 65, 95, 127):
        W = L.orc_words_for_k(k)
        for _ in range(200):
            x = L.orc_kmer_from_str(_rand_kmer(rng, k).encode(), k)
            iv = int(rng.integers(0, 2**32))
            a = np.array(_words(x, W), dtype=np.uint64)
            assert L.orc_kmer_hash(x, k, iv) == R.ref_lk3_hashlittle(a.ctypes.data, 8 * W, iv)


@pytest.mark.parametrize("k", [3, 15, 29, 31, 33, 47, 63, 65, 79, 95, 97, 111, 127])
def test_hash_matches_reference_kmer_hash_h(orc, mcx, k):
    """The reference's own BinaryKmer hash -- bklk3_hashlittle, src/kmer/kmer_hash.h:162-211, compiled unmodified
    into oracle/_ref with NUM_BKMER_WORDS = 1 .. 4 -- against the oracle's orc_kmer_hash and the product's host
    template mcx_kmer_hash (the device uses the same template: mcx_kmer.h kmer_hash<W>), on canonical keys and
    arbitrary seeds (hash_table.c:132-145 rehashes with seed + i)."""
    L = orc.lib()
    W = L.orc_words_for_k(k)
    R = orc.ref_revcmp(W)
    if R is None or not hasattr(R, "ref_bklk3_hashlittle"):
        pytest.skip("oracle/_ref/librevcmp%d.so with kmer_hash.h not built (needs /root/reference at build time)" % W)
    rng = np.random.default_rng(500 + k)
    for s in [_rand_kmer(rng, k) for _ in range(300)] + ["A" * k, "T" * k, "ACG" * (k // 3) + "A" * (k % 3)]:
        x = L.orc_kmer_from_str(s.encode(), k)
        key = L.orc_kmer_get_key(x, k)
        a = np.array(_words(key, W), dtype=np.uint64)
        for iv in (0, 1, 7, int(rng.integers(0, 2**32))):
            want = R.ref_bklk3_hashlittle(a.ctypes.data, iv)
            assert L.orc_kmer_hash(key, k, iv) == want
            assert mcx.kmer_hash([int(w) for w in a], k, iv) == want


@pytest.mark.parametrize("k", list(range(3, 128, 2)))
def test_revcomp_matches_reference_revcmp(orc, mcx, k):
    """orc_kmer_revcomp / orc_kmer_get_key and the product's host template mcx_kmer_canonical against
    the reference's own code: dev/bkmer_revcmp/revcmp.c (all four binary_kmer_reverse_complement{1..4};
    number 2 is src/graph/binary_kmer.c:102-133 line for line), compiled unmodified with
    NUM_BKMER_WORDS = 1 .. 4 into oracle/_ref.  Pins the b[0]-is-the-top-word layout, the base order
    inside a word and the shift across words for every odd k up to 127 (mccortex95 / mccortex127)."""
    L = orc.lib()
    W = L.orc_words_for_k(k)
    R = orc.ref_revcmp(W)
    if R is None:
        pytest.skip("oracle/_ref/librevcmp%d.so not built (needs /root/reference at build time)" % W)
    rng = np.random.default_rng(100 + k)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    buf = C.create_string_buffer(160)
    cases = [_rand_kmer(rng, k) for _ in range(60)] + ["A" * k, "T" * k, "C" * k, "G" * k, "AC" * (k // 2) + "G"]
    for s in cases:
        x = L.orc_kmer_from_str(s.encode(), k)
        a = np.array(_words(x, W), dtype=np.uint64)
        want = None
        for method in (1, 2, 3, 4):
            out = np.zeros(W, dtype=np.uint64)
            R.ref_revcmp(method, a.ctypes.data, k, out.ctypes.data)
            got = [int(w) for w in out]
            assert want is None or got == want  # the reference's four variants agree with each other
            want = got
        rc = L.orc_kmer_revcomp(x, k)
        assert _words(rc, W) == want
        # ... and the words the reference code produced spell the reverse complement
        y = orc.BKmer()
        for i in range(W):
            y.b[i] = want[i]
        L.orc_kmer_to_str(y, k, buf)
        assert buf.value.decode() == "".join(comp[c] for c in reversed(s))
        # canonical key = the smaller of the two, word 0 (the top word) first (binary_kmer.c:43-57)
        fw = _words(x, W)
        key = min(fw, want)
        assert _words(L.orc_kmer_get_key(x, k), W) == key
        kw, o = mcx.kmer_canonical(fw, k)
        assert kw == key and o == (0 if key == fw else 1)


def test_hash_table_cap_kats(orc):
    L = orc.lib()
    for n, nb, bs in KATS["hash_table_cap"]:
        b, s = C.c_uint64(), C.c_uint8()
        assert L.orc_hash_table_cap(n, C.byref(b), C.byref(s)) == nb * bs
        assert (b.value, s.value) == (nb, bs)


@pytest.mark.parametrize("k", [3, 5, 19, 31, 33, 39, 63, 95, 127])
def test_bkmer_properties(orc, k):
    """src/tests/bkmer_tests.c: str<->bin round trip (:5-25), revcmp involution and
    revcmp(x) != x for odd k (:27-46), shift identities (:48-126), first/last nuc (:128-150)."""
    L = orc.lib()
    W = L.orc_words_for_k(k)
    rng = np.random.default_rng(k)
    buf = C.create_string_buffer(256)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(100):
        s = _rand_kmer(rng, k)
        x = L.orc_kmer_from_str(s.encode(), k)
        L.orc_kmer_to_str(x, k, buf)
        assert buf.value.decode() == s
        rc = L.orc_kmer_revcomp(x, k)
        L.orc_kmer_to_str(rc, k, buf)
        assert buf.value.decode() == "".join(comp[c] for c in reversed(s))
        assert _words(L.orc_kmer_revcomp(rc, k), W) == _words(x, W)
        assert _words(rc, W) != _words(x, W)
        # rolling == from_str of the shifted string
        n = int(rng.integers(0, 4))
        y = L.orc_kmer_shift_add(x, k, n)
        assert _words(y, W) == _words(L.orc_kmer_from_str((s[1:] + "ACGT"[n]).encode(), k), W)
        # key is the smaller of the two strands, compared word 0 first
        key = L.orc_kmer_get_key(x, k)
        assert _words(key, W) == min(_words(x, W), _words(rc, W))
        # top word never uses more than 2*(k&31) bits
        assert x.b[0] >> (2 * (k & 31)) == 0


def test_contig_split_kat(orc):
    L = orc.lib()
    kat = KATS["contig_split_k31"]
    rng = np.random.default_rng(0)
    r = list(_rand_kmer(rng, kat["length"]))
    for p in kat["n_positions"]:
        r[p] = "N"
    r = "".join(r).encode()
    ss = C.c_size_t()
    got, search = [], 0
    while True:
        cs = L.orc_contig_start(r, len(r), None, 0, search, 31, 0, 0)
        if cs >= len(r):
            break
        ce = L.orc_contig_end(r, len(r), None, 0, cs, 31, 0, 0, C.byref(ss))
        got.append([cs, ce])
        search = ss.value
    assert got == kat["contigs"]


def test_reference_build_graph_kat(orc):
    kat = KATS["build_graph_tests_k19"]
    g = orc.Graph(19, 1, 1024)
    reads = kat["reads"]
    tot = orc.Stats()
    for i in range(3):
        b, o = orc.pack_reads(reads[2 * i:2 * i + 2])
        g.add_reads(0, b, o, stats=tot)
        for kmer, cov in kat["covg_after_each_pair"].items():
            assert g.lookup(kmer)[0][0] == cov[i]
    g.update_stats(0, tot)
    assert tot.total_bases_loaded == kat["total_sequence"] and tot.contigs_parsed == kat["contigs"]
    ctx = g.ctx_bytes(True)
    mean, total = np.frombuffer(ctx[22:26], np.uint32)[0], np.frombuffer(ctx[26:34], np.uint64)[0]
    assert total == 168 and abs(float(mean) - (168 / 6 + 0.5)) <= 0.5  # build_graph_tests.c:135-148


def test_reference_pcr_duplicate_unit_test(orc):
    """src/tests/build_graph_tests.c:19-148 replayed call by call through the restatement"""
    import pcr_cases as pc
    g = orc.Graph(pc.K, 1, 1024)
    tot = orc.Stats()
    dup_se = dup_pe = 0
    for r1, r2, md, pcr, (c1, c2) in pc.STEPS:
        b, o = orc.pack_reads([r1] if r2 is None else [r1, r2])
        if pcr:
            _, (dse, dpe, npe) = g.add_reads_pcr(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF,
                                                 paired=r2 is not None, matedir=md, stats=tot)
            dup_se += dse
            dup_pe += dpe
        else:
            g.add_reads(0, b, o, fq_cutoff=pc.FQ_CUTOFF, hp_cutoff=pc.HP_CUTOFF, stats=tot)
        for kmer, c in ((pc.K1, c1), (pc.K2, c2)):
            if c is not None:
                assert g.lookup(kmer)[0][0] == c, (r1, r2, md)
    assert (dup_se, dup_pe) == (2, 6)  # the empty pair counts as a duplicate too (build_graph.c:78-84)
    assert tot.total_bases_loaded == pc.TOTAL_SEQ and tot.contigs_parsed == pc.CONTIGS
    g.update_stats(0, tot)
    ctx = g.ctx_bytes(True)
    mean, total = np.frombuffer(ctx[22:26], np.uint32)[0], np.frombuffer(ctx[26:34], np.uint64)[0]
    assert total == pc.TOTAL_SEQ and abs(float(mean) - (pc.TOTAL_SEQ / pc.CONTIGS + 0.5)) <= 0.5  # :135-148


def test_pcr_filter_order_and_reset(orc):
    """first read at a start wins; the other strand is a different start; reset forgets (ctx_build.c:392-395)"""
    g = orc.Graph(5, 2, 1024)
    reads = ["ACGTTGCA", "ACGTTCCC", "TGCAACGT", "ACGTTGGG"]  # 0, 1, 3 start at ACGTT forward; 2 = revcomp of 0
    b, o = orc.pack_reads(reads)
    st, (dse, dpe, npe) = g.add_reads_pcr(0, b, o, matedir="FF")
    assert (dse, dpe, npe) == (2, 0, 0) and st.num_se_reads == 4 and st.num_good_reads == 2
    assert g.lookup("CGTTC") is None                   # read 1 was dropped
    assert g.lookup("ACGTT")[0][0] == 2 and g.lookup("TGCAA")[0][0] == 2  # read 0 and its reverse complement
    _, (dse, dpe, npe) = g.add_reads_pcr(0, b, o, matedir="FF")
    assert dse == 4                                   # every start is taken now
    g.pcr_reset()
    _, (dse, dpe, npe) = g.add_reads_pcr(1, b, o, matedir="FF")
    assert dse == 2 and g.lookup("ACGTT")[0][1] == 2


def test_header_layout_kats(orc):
    g = orc.Graph(31, 1, 1024)
    g.set_sample(0, "abcde")
    b, o = orc.pack_reads(["ACGTTGCATGCATGCAAGTCCGATAGCTAGCT"])
    st = g.add_reads(0, b, o)
    g.update_stats(0, st)
    ctx = g.ctx_bytes(True)
    assert g.header_size() == KATS["header_1col_5char_name_bytes"]
    assert len(ctx) == 90 + 13 * 2
    assert ctx[:6] == b"CORTEX" and ctx[84:90] == b"CORTEX"
    assert np.frombuffer(ctx[6:22], np.uint32).tolist() == [6, 31, 1, 1]
    assert ctx[43:59].hex() == KATS["seq_err_bytes"]
    # first k-mer of the read: only the 'next' side is set ('.......T')
    cov, edg = g.lookup(KATS["edge_first_kmer_of_read"]["kmer"])
    assert cov[0] == 1 and edg[0] == 0x08
    g2 = orc.Graph(31, 2, 1024)
    g2.set_sample(0, "alice"); g2.set_sample(1, "bob")
    assert g2.header_size() == KATS["header_2col_alice_bob_bytes"]
    for (k, nc), rs in {(31, 1): 13, (31, 2): 18, (63, 2): 26, (63, 1): 21, (31, 4): 28}.items():  # W*8 + 5*cols (graph_writer.c:121; SURVEY 8c prints 17 for (31,2): a typo, 8+2*4+2=18)
        gg = orc.Graph(k, nc, 1024)
        assert (gg.ctx_bytes().__len__() - gg.header_size()) == 0
        assert 8 * gg.W + 5 * nc == rs


def test_sorted_ctx_is_seed_and_thread_independent(orc):
    """the sorted .ctx depends only on the input (tests/sort/Makefile:29-45)."""
    bases, offs = synth.reads(3000, 100, genome_len=20000, seed=9, n_frac=0.05)
    ref = None
    for seed, nt in [(1, 1), (77777, 1), (5, 4), (123, 8)]:
        g = orc.Graph(31, 1, 1 << 18, seed=seed)
        st = g.add_reads(0, bases, offs, nthreads=nt)
        g.update_stats(0, st)
        ctx = g.ctx_bytes(True)
        if ref is None:
            ref = ctx
            unsorted = g.ctx_bytes(False)
            assert len(unsorted) == len(ctx)
        assert ctx == ref


def test_mt_insert_each_key_novel_once(orc):
    """src/tests/hash_table_tests.c:80-129: many threads insert the same keys; every key is
    novel exactly once."""
    bases, offs = synth.reads(4000, 100, genome_len=30000, seed=4)
    g1 = orc.Graph(31, 1, 1 << 18)
    s1 = g1.add_reads(0, bases, offs, nthreads=1)
    g8 = orc.Graph(31, 1, 1 << 18)
    s8 = g8.add_reads(0, bases, offs, nthreads=8)
    assert s8.num_kmers_novel == s1.num_kmers_novel == g1.nkmers == g8.nkmers


@pytest.mark.parametrize("k", [21, 31, 45, 63])
def test_table_graph_equals_tuple_reduction(orc, k):
    """The reference-shaped table build (edges read back from stored keys, db_graph.c:152-166)
    equals the order-independent reduction of per-occurrence tuples."""
    bases, offs = synth.reads(2000, 90, genome_len=8000, seed=k, n_frac=0.1, lower_frac=0.2, var_len=True)
    g = orc.Graph(k, 1, 1 << 18)
    g.add_reads(0, bases, offs)
    keys, edges = orc.tuples(k, bases, offs)
    K, Cv, E = orc.graph_from_tuples(keys, edges)
    W = g.W
    rec = np.frombuffer(g.ctx_bytes(True)[g.header_size():], np.uint8).reshape(-1, 8 * W + 5)
    assert (rec[:, :8 * W].copy().view(np.uint64).reshape(-1, W) == K).all()
    assert (rec[:, 8 * W:8 * W + 4].copy().view(np.uint32).ravel() == Cv[:, 0]).all()
    assert (rec[:, 8 * W + 4] == E[:, 0]).all()


def test_quality_homopolymer_rules(orc):
    """seq_reader.c:61-172 corner cases: asymmetric '>' / '<' on the quality cutoff and the
    homopolymer search restart."""
    L = orc.lib()
    ss = C.c_size_t()
    seq = b"ACGTACGTACGTACGTACGT"
    q = bytes([40] * 8 + [30] + [40] * 11)
    # start needs qual > cutoff for all k bases: cutoff 30 rejects windows holding position 8
    assert L.orc_contig_start(seq, 20, q, 20, 0, 5, 30, 0) == 0
    assert L.orc_contig_end(seq, 20, q, 20, 0, 5, 30, 0, C.byref(ss)) == 20  # extension stops only at qual < cutoff
    assert L.orc_contig_end(seq, 20, q, 20, 0, 5, 31, 0, C.byref(ss)) == 8
    assert L.orc_contig_start(seq, 20, q, 20, 8, 5, 31, 0) == 9
    hp = b"ACGTAAAAAACGTACGT"
    assert L.orc_contig_start(hp, len(hp), None, 0, 0, 5, 0, 4) == 0
    e = L.orc_contig_end(hp, len(hp), None, 0, 0, 5, 0, 4, C.byref(ss))
    assert e == 7 and ss.value == 4  # run A A A (A) would reach 4 at index 7


def test_w1_fast_path_equals_generic_code(orc):
    """The single-word fast path used for the timed CPU baseline gives the same graph as the
    generic multi-word restatement."""
    bases, offs = synth.reads(3000, 100, genome_len=15000, seed=12, n_frac=0.1, lower_frac=0.2)
    for k in (5, 21, 31):
        a = orc.Graph(k, 2, 1 << 18)
        b = orc.Graph(k, 2, 1 << 18)
        b.force_generic()
        for g in (a, b):
            g.add_reads(0, bases, offs)
            g.add_reads(1, bases[:int(offs[1000])], offs[:1001], nthreads=4)
        assert a.ctx_bytes(True) == b.ctx_bytes(True)


def test_timed_baseline_helpers_build_the_same_graph(orc, tmp_path):
    """orc_graph_tune (prefault, per-worker node tallies) and orc_build_file (1 reader thread ->
    2048-slot pool -> workers: async_read_io.c:145-175,283-310) are bench.py's CPU baseline; the
    graphs they build are the plain oracle's."""
    import synth
    bases, offs = synth.reads(3000, 120, genome_len=20000, seed=5, n_frac=0.05)
    ref = orc.Graph(31, 1, 1 << 20)
    st0 = ref.add_reads(0, bases, offs)
    want = ref.ctx_bytes(True)
    g = orc.Graph(31, 1, 1 << 20)
    g.tune(3)
    st = g.add_reads(0, bases, offs, nthreads=3)
    assert g.nkmers == ref.nkmers and g.ctx_bytes(True) == want
    assert st.as_dict() == st0.as_dict()
    fq = tmp_path / "r.fq"
    with open(fq, "wb") as f:
        for i in range(len(offs) - 1):
            r = bytes(bases[int(offs[i]):int(offs[i + 1])])
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    g2 = orc.Graph(31, 1, 1 << 20)
    g2.tune(2)
    tot, ins, st2 = g2.build_file(str(fq), 4)
    assert tot >= ins > 0 and g2.nkmers == ref.nkmers and g2.ctx_bytes(True) == want
    assert st2.num_kmers_loaded == st0.num_kmers_loaded and st2.contigs_parsed == st0.contigs_parsed
