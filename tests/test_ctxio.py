"""`.ctx` reading, colour filters, GraphInfo merging and `build --graph`.

CPU: the Python restatement (oracle/ctxio.py) is pinned to the C oracle's header writer (itself
pinned in tests/test_oracle.py) and to the golden files; the host program's argument handling.
GPU: mcx_graph_add_records against the oracle's graph_load restatement, and the command line."""
import os
import struct
import subprocess

import numpy as np
import pytest

import synth
from oracle import ctxio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mccortex_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(maxk, *args, stdin=None):
    exe = os.path.join(BIN, "mccortex%d" % maxk)
    p = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, input=stdin)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


def _reads_graph(orc, k, ncols, jobs, names):
    """C oracle graph + the GraphInfo the reference would hold after build_graph()"""
    og = orc.Graph(k, ncols, 1 << 20)
    gi = [ctxio.GraphInfo() for _ in range(ncols)]
    for c, n in enumerate(names):
        og.set_sample(c, n)
        gi[c].sample_name = n
    for col, b, o in jobs:
        st = og.add_reads(col, b, o)
        og.update_stats(col, st)
        gi[col].update_contigs(st.total_bases_loaded, st.contigs_parsed)
    return og, gi


@pytest.mark.parametrize("k", [31, 63, 5])
def test_python_header_equals_c_oracle_header(orc, k):
    g = synth.genome(8000, 4)
    jobs = [(0, *synth.reads(500, 90, seed=1, g=g, n_frac=0.1)), (1, *synth.reads(300, 120, seed=2, g=g, var_len=True)),
            (0, *synth.reads(200, 70, seed=3, g=g))]
    og, gi = _reads_graph(orc, k, 3, jobs, ["alice", "bob"])   # colour 2 stays empty / "undefined"
    want = og.ctx_bytes(True)
    hs = og.header_size()
    assert ctxio.header_bytes(k, gi) == want[:hs]
    hdr, n = ctxio.read_header(want)
    assert n == hs and hdr["kmer_size"] == k and hdr["num_cols"] == 3
    assert [x.sample_name for x in hdr["ginfo"]] == ["alice", "bob", "undefined"]
    # (writing the parsed header again is NOT a fixed point: graph_info_merge re-derives the mean
    # read length from total / round(total / mean) every time a header passes through it)
    keys, covgs, edges = ctxio.records(want, hdr, hs)
    assert len(keys) == og.nkmers and covgs.shape == (og.nkmers, 3)


@pytest.mark.parametrize("name", ["tiny_k31", "tiny_k63", "tiny_k5"])
def test_golden_files_parse(name):
    buf = open(os.path.join(GOLD, name + ".ctx"), "rb").read()
    hdr, hs = ctxio.read_header(buf)
    keys, covgs, edges = ctxio.records(buf, hdr, hs)
    assert hs + len(keys) * (8 * hdr["num_words"] + 5 * hdr["num_cols"]) == len(buf)
    assert (covgs.sum(axis=1) > 0).all()
    k = keys.astype(object)  # sorted by key, most significant word first
    as_int = [sum(int(w) << (64 * (hdr["num_words"] - 1 - i)) for i, w in enumerate(row)) for row in k]
    assert as_int == sorted(as_int)


def test_filter_parsing():
    pf = ctxio.parse_filter
    assert pf("in.ctx", 3, 0) == ("in.ctx", [(0, 0), (1, 1), (2, 2)])
    assert pf("in.ctx", 2, 4) == ("in.ctx", [(0, 4), (1, 5)])
    assert pf("in.ctx:0,6-8", 9, 1) == ("in.ctx", [(0, 1), (6, 2), (7, 3), (8, 4)])
    assert pf("2:in.ctx:1", 3, 0) == ("in.ctx", [(1, 2)])
    assert pf("0:in.ctx", 3, 7) == ("in.ctx", [(0, 0), (1, 0), (2, 0)])      # flatten into colour 0
    assert pf("3,1:dir/in.ctx:0-1", 2, 0) == ("dir/in.ctx", [(1, 1), (0, 3)])
    assert pf("in.ctx:2-0", 3, 0) == ("in.ctx", [(2, 0), (1, 1), (0, 2)])
    assert pf("a:b.ctx", 1, 0) == ("a:b.ctx", [(0, 0)])                      # not a range: part of the path
    for bad in ["in.ctx:3", "in.ctx:0,", "0,1,2:in.ctx:0-1"]:
        with pytest.raises(ctxio.CtxError):
            pf(bad, 3, 0)


def test_graphinfo_merge_arithmetic():
    a, b = ctxio.GraphInfo(), ctxio.GraphInfo()
    a.sample_name, a.total_sequence, a.mean_read_length = "x", 1000, 100
    b.sample_name, b.total_sequence, b.mean_read_length = "y", 3000, 150
    b.seq_err = ctxio.LD(0.02)
    b.cleaning.cleaned_kmers, b.cleaning.clean_kmers_thresh = 1, 5
    b.cleaning.is_graph_intersection, b.cleaning.intersection_name = 1, "ref"
    a.merge(b)
    assert a.sample_name == "x,y" and a.total_sequence == 4000
    assert a.mean_read_length == 4000 // (10 + 20)
    assert abs(float(a.seq_err) - (0.01 * 1000 + 0.02 * 3000) / 4000) < 1e-15
    assert a.cleaning.cleaned_kmers == 1 and a.cleaning.clean_kmers_thresh == 5
    assert a.cleaning.is_graph_intersection == 1 and a.cleaning.intersection_name == "ref"
    c = ctxio.GraphInfo()
    c.cleaning.is_graph_intersection, c.cleaning.intersection_name = 1, "other"
    a.merge(c)
    assert a.cleaning.intersection_name == "ref,other" and a.sample_name == "x,y"
    # header fields survive a write/read round trip
    hdr, _ = ctxio.read_header(ctxio.header_bytes(31, [a]))
    g = hdr["ginfo"][0]
    assert (g.sample_name, g.total_sequence, g.cleaning.clean_kmers_thresh, g.cleaning.intersection_name) == \
        ("x,y", 4000, 5, "ref,other")
    assert g.seq_err == a.seq_err


def test_oracle_add_record_semantics(orc):
    og = orc.Graph(31, 2, 1 << 12)
    key = [0x06f939390b58c9c9]
    assert og.add_record(key, [0, 0], [3, 0]) == 0 and og.nkmers == 0          # no coverage: not loaded
    assert og.add_record(key, [5, 0], [0x11, 0], must_exist=True) == 0 and og.nkmers == 0
    assert og.add_record(key, [5, 0], [0x11, 0]) == 1 and og.nkmers == 1
    assert og.add_record(key, [0xFFFFFFFF, 2], [0x02, 0x80], must_exist=True) == 1 and og.nkmers == 1
    body = og.body_bytes()
    assert body == struct.pack("<QIIBB", key[0], 0xFFFFFFFF, 2, 0x13, 0x80)     # saturating sum, edges OR


def test_graph_option_argument_errors(mcx, tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_text(">r\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    good = os.path.join(GOLD, "tiny_k31.ctx")
    bad = tmp_path / "bad.ctx"
    bad.write_bytes(b"CORTEY" + open(good, "rb").read()[6:])
    trunc = tmp_path / "trunc.ctx"
    trunc.write_bytes(open(good, "rb").read()[:40])
    cases = [
        (["build", "-k", "31", "-g", good, "o.ctx"], "No inputs given"),
        (["build", "-k", "31", "-g", str(tmp_path / "none.ctx"), "-s", "a", "--seq", str(fa), "o.ctx"], "Cannot open file"),
        (["build", "-k", "31", "-g", str(bad), "-s", "a", "--seq", str(fa), "o.ctx"], "Magic word doesn't match"),
        (["build", "-k", "31", "-g", str(trunc), "-s", "a", "--seq", str(fa), "o.ctx"], "Unexpected end of file"),
        (["build", "-k", "21", "-g", good, "-s", "a", "--seq", str(fa), "o.ctx"], "Input graph kmer_size doesn't match"),
        (["build", "-k", "31", "-g", good + ":7", "-s", "a", "--seq", str(fa), "o.ctx"], "Invalid filter path"),
        (["build", "-k", "31", "-g", os.path.join(GOLD, "tiny_k63.ctx"), "-s", "a", "--seq", str(fa), "o.ctx"], "Cannot handle kmer size 63"),
    ]
    for args, msg in cases:
        rc, _, err = run(31, *args)
        assert rc == 1, (args, err)
        assert msg in err, (args, err)


def _probe(tmp_path):
    """tests/ctx_filter_probe.c over the host program's own reader (no GPU needed)"""
    exe = tmp_path / "ctx_filter_probe"
    host = os.path.join(ROOT, "mccortex_amd", "host")
    subprocess.check_call(["gcc", "-O1", "-I", host, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "ctx_filter_probe.c"), os.path.join(host, "ctx_file.c"),
                           os.path.join(host, "host_util.c"), "-o", str(exe), "-lm"])

    def call(arg, into_offset=0):
        p = subprocess.run([str(exe), arg, str(into_offset)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            return None, p.stderr.decode(errors="replace")
        out = {"filter": [], "colour": []}
        for line in p.stdout.decode().splitlines():
            key, _, rest = line.partition(" ")
            if key in ("filter",):
                out["filter"].append(tuple(int(x) for x in rest.split()))
            elif key == "colour":
                out["colour"].append(rest.split("|"))
            else:
                out[key] = rest
        return out, ""
    return call


def test_host_reader_filters_match_the_checker(tmp_path, monkeypatch):
    """The C host program's "<into>:path:<from>" handling against oracle/ctxio.parse_filter: the
    reference's documented cases, the edge cases of its list syntax, and random arguments."""
    call = _probe(tmp_path)
    ncols = 9
    gis = [ctxio.GraphInfo() for _ in range(ncols)]
    for c, g in enumerate(gis):
        g.sample_name, g.total_sequence, g.mean_read_length = "s%d" % c, 1000 * c, 100 + c
    gis[3].cleaning.cleaned_kmers, gis[3].cleaning.clean_kmers_thresh = 1, 7
    gis[4].cleaning.is_graph_intersection, gis[4].cleaning.intersection_name = 1, "pop"
    f = tmp_path / "nine.ctx"
    hdr = ctxio.header_bytes(31, gis)
    f.write_bytes(hdr + bytes(3 * (8 + 5 * ncols)))
    plain, _ = call(str(f), 2)
    assert plain["path"] == str(f) and plain["dims"] == "6 31 1 9 %d 3" % len(hdr)
    assert plain["filter"] == [(c, c + 2) for c in range(ncols)] and plain["into_ncols"] == "11"
    assert [c[1] for c in plain["colour"]] == ["s%d" % c for c in range(ncols)]
    assert plain["colour"][3][5:8] == ["0010", "0", "7"] and plain["colour"][4][5] == "0001" and plain["colour"][4][8] == "pop"
    assert [int(c[3]) for c in plain["colour"]] == [1000 * c for c in range(ncols)]

    def same(arg, off=0):
        got, err = call(arg, off)
        try:
            path, want = ctxio.parse_filter(arg, ncols, off)
        except ctxio.CtxError:
            assert got is None and "Invalid filter path" in err, (arg, got, err)
            return False
        assert got is not None, (arg, err)
        assert got["path"] == path and got["filter"] == want, (arg, got, want)
        assert int(got["into_ncols"]) == max(i for _, i in want) + 1
        return True

    p = str(f)
    for arg in [p, p + ":0,6-8", "2:" + p + ":1", "0:" + p, "3,1:" + p + ":0-1", p + ":2-0", p + ":", p + ":8", "7-4:" + p + ":0-3",
                "0-8:" + p, "12:" + p + ":5,5,5", p + ":0-0", p + ":3-3,4"]:
        assert same(arg, 1), arg
    for arg in [p + ":9", p + ":0,", p + ":,0", p + ":0,,1", p + ":1-", p + ":-1", p + ":1-2-3", "0,1,2:" + p + ":0-1", "0-1:" + p + ":0-2",
                p + ":0-9", "1-:" + p, ",:" + p]:
        assert not same(arg), arg
    # a leading list that is not closed by ':' and a trailing one that is not opened by one belong to the path
    odd = tmp_path / "7-8"
    odd.write_bytes(f.read_bytes())
    monkeypatch.chdir(tmp_path)
    assert same("7-8") and same("./7-8:2") and same("1:./7-8:2") and same("3:./7-8")
    rng = np.random.default_rng(5)
    items = lambda n: ",".join(("%d" % a if a == b else "%d-%d" % (a, b))
                               for a, b in zip(rng.integers(0, n, 3), rng.integers(0, n, 3)))[: int(rng.integers(1, 12))]
    ok = 0
    for _ in range(150):
        arg = p
        if rng.random() < 0.6:
            arg = arg + ":" + items(10)
        if rng.random() < 0.5:
            arg = items(14) + ":" + arg
        ok += same(arg, int(rng.integers(0, 4)))
    assert 20 < ok < 140


# ------------------------------------------------------------------------------------------
def _random_records(rng, k, file_ncols, n, W):
    keys = np.zeros((n, W), dtype=np.uint64)
    top_bits = 2 * k - 64 * (W - 1)
    keys[:, 0] = rng.integers(0, 1 << min(top_bits, 62), n, dtype=np.uint64)
    if top_bits > 62:
        keys[:, 0] |= rng.integers(0, 1 << (top_bits - 62), n, dtype=np.uint64) << np.uint64(62)
    for w in range(1, W):
        keys[:, w] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    keys[n // 2:] = keys[:n - n // 2]          # duplicates: coverage sums, edges OR
    keys[0] = 0                                # the all-A k-mer
    covgs = rng.integers(0, 4, (n, file_ncols), dtype=np.uint32) * rng.integers(0, 1 << 30, (n, file_ncols), dtype=np.uint32)
    covgs[1::7] = 0xFFFFFFFF                   # saturation
    edges = rng.integers(0, 256, (n, file_ncols), dtype=np.uint8)
    edges[covgs == 0] = 0                      # well-formed: no edges without coverage
    rec = np.concatenate([keys.view(np.uint8).reshape(n, 8 * W), covgs.view(np.uint8).reshape(n, 4 * file_ncols), edges], axis=1)
    return keys, covgs, edges, rec.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("k,file_ncols,ncols,filt", [
    (31, 1, 1, [(0, 0)]),
    (31, 3, 2, [(0, 1), (2, 1), (1, 0)]),
    (63, 2, 3, [(1, 0), (1, 2)]),
    (45, 4, 2, [(3, 1)]),
])
def test_add_records_matches_oracle(mcx, orc, k, file_ncols, ncols, filt):
    rng = np.random.default_rng(k * 10 + file_ncols)
    og = orc.Graph(k, ncols, 1 << 20)
    g = mcx.Graph(k, ncols, 1 << 20)
    # some reads first (both insert strategies meet the loaded records), then records, then reads
    b, o = synth.reads(2000, 100, seed=5, genome_len=20000)
    og.add_reads(0, b, o); g.add_reads(0, b, o)
    keys, covgs, edges, rec = _random_records(rng, k, file_ncols, 30000, og.W)
    # records of k-mers the reads created: take some keys back from the graph
    kk, _, _ = g.records(True)
    nk = min(len(kk), 5000)
    keys[100:100 + nk] = kk[:nk]
    rec = np.concatenate([keys.view(np.uint8).reshape(len(keys), 8 * og.W), covgs.view(np.uint8).reshape(len(keys), 4 * file_ncols),
                          edges], axis=1).tobytes()
    loaded = 0
    for i in range(len(keys)):
        cv, ed = [0] * ncols, [0] * ncols
        for f, t in filt:
            cv[t] = min(0xFFFFFFFF, cv[t] + int(covgs[i, f])); ed[t] |= int(edges[i, f])
        loaded += og.add_record(keys[i], cv, ed) == 1
    st = g.add_records(rec, file_ncols, filt)
    assert (st.nkmers_read, st.nkmers_loaded) == (len(keys), loaded)
    assert st.first_oversized == -1 and st.first_edges_no_covg == -1
    b2, o2 = synth.reads(1500, 80, seed=6, genome_len=20000)
    og.add_reads(ncols - 1, b2, o2); g.add_reads(ncols - 1, b2, o2)
    assert g.nkmers == og.nkmers
    assert g.export(True) == og.body_bytes(True)
    # must_exist: nothing new appears, existing records are updated
    st2 = g.add_records(rec, file_ncols, filt, must_exist=True)
    for i in range(len(keys)):
        cv, ed = [0] * ncols, [0] * ncols
        for f, t in filt:
            cv[t] = min(0xFFFFFFFF, cv[t] + int(covgs[i, f])); ed[t] |= int(edges[i, f])
        og.add_record(keys[i], cv, ed, must_exist=True)
    assert st2.nkmers_novel == 0 and g.nkmers == og.nkmers
    assert g.export(True) == og.body_bytes(True)
    g.close()


@pytest.mark.gpu
def test_add_records_reports_dirty_records(mcx):
    g = mcx.Graph(31, 1, 1 << 12)
    rec = struct.pack("<QIB", 5, 0, 0) + struct.pack("<QIB", 6, 0, 4) + struct.pack("<QIB", 7, 2, 1)
    st = g.add_records(rec, 1, [(0, 0)])
    assert (st.nkmers_read, st.nkmers_loaded, st.first_zero_covg, st.first_edges_no_covg) == (3, 1, 0, 1)
    with pytest.raises(mcx.McxError):
        g.add_records(struct.pack("<QIB", 1 << 62, 1, 0), 1, [(0, 0)])   # oversized for k=31
    with pytest.raises(mcx.McxError):
        g.add_records(rec, 1, [(1, 0)])
    g.close()


def _fasta(path, bases, offs):
    with open(path, "w") as f:
        for i in range(len(offs) - 1):
            f.write(">r%d\n%s\n" % (i, bytes(bases[int(offs[i]):int(offs[i + 1])]).decode()))
    return str(path)


@pytest.mark.gpu
@pytest.mark.parametrize("maxk,k", [(31, 31), (63, 47)])
def test_build_graph_option_matches_oracle(mcx, orc, tmp_path, maxk, k):
    g0 = synth.genome(30000, 8)
    r = [synth.reads(1500, 100, seed=20 + i, g=g0, n_frac=0.05) for i in range(4)]
    f = [_fasta(tmp_path / ("r%d.fa" % i), *r[i]) for i in range(4)]
    # A: 2 colours from reads (this command line is covered by tests/test_cli.py)
    A = str(tmp_path / "A.ctx")
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "--sort", "-s", "a0", "--seq", f[0], "-s", "a1", "--seq", f[1], A)
    assert rc == 0, err
    ogA, giA = _reads_graph(orc, k, 2, [(0, *r[0]), (1, *r[1])], ["a0", "a1"])
    bufA = open(A, "rb").read()
    assert bufA == ctxio.header_bytes(k, giA) + ogA.body_bytes(True)

    def expect(ncols, loads, names, jobs):
        """loads: [(buf, input spec, into_offset)] in command-line order; names {colour: name};
        jobs: [(colour, bases, offs)] -> expected file bytes (ctx_build.c:365-425)"""
        og = orc.Graph(k, ncols, 1 << 20)
        gi = [ctxio.GraphInfo() for _ in range(ncols)]
        for buf, spec, off in loads:
            hdr, _ = ctxio.read_header(buf)
            _, filt = ctxio.parse_filter(spec, hdr["num_cols"], off)
            ctxio.load_into(og, gi, buf, filt)
        for c, n in names.items():
            gi[c].sample_name = n        # strbuf_set: overrides what the graph files merged in
        for col, b, o in jobs:
            st = og.add_reads(col, b, o)
            gi[col].update_contigs(st.total_bases_loaded, st.contigs_parsed)
        return ctxio.header_bytes(k, gi) + og.body_bytes(True)

    # 1. graph into colours 0-1, new sample in colour 2
    out = str(tmp_path / "o1.ctx")
    rc, _, err = run(maxk, "build", "-k", str(k), "-n", "1M", "--sort", "-g", A, "-s", "new", "--seq", f[2], out)
    assert rc == 0, err
    assert "[GReader] Loaded" in err and "kmers parsed" in err
    assert open(out, "rb").read() == expect(3, [(bufA, A, 0)], {2: "new"}, [(2, *r[2])])

    # 2. sample first, then a filtered graph (colour 1 of A) after it, then another sample; via stdin too
    out = str(tmp_path / "o2.ctx")
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "--sort", "-s", "s0", "--seq", f[2], "-g", A + ":1",
                     "-s", "s1", "--seq", f[3], out)
    assert rc == 0, err
    # -s s0 -> colour 0; -g with into_offset 0 loads A:1 into colour 0 as well; -s s1 -> colour 1
    assert open(out, "rb").read() == expect(2, [(bufA, A + ":1", 0)], {0: "s0", 1: "s1"}, [(0, *r[2]), (1, *r[3])])

    # 3. explicit into-colours, both file colours flattened into colour 1; graph read from stdin
    out = str(tmp_path / "o3.ctx")
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "--sort", "-g", "1:-", "-s", "x", "--seq", f[3], out, stdin=bufA)
    assert rc == 0, err
    assert open(out, "rb").read() == expect(3, [(bufA, "1:-", 0)], {2: "x"}, [(2, *r[3])])

    # 4. two graph files: the second lands after the first's colours; loading our own output again
    out4 = str(tmp_path / "o4.ctx")
    o1 = open(str(tmp_path / "o1.ctx"), "rb").read()
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "--sort", "-g", A, "-g", str(tmp_path / "o1.ctx") + ":2,0",
                     "-s", "z", "--seq", f[0], out4)
    assert rc == 0, err
    # first -g: intocolour 0 -> 1; second -g is opened with into_offset 1 (colours 1, 2); -s z -> colour 3
    assert open(out4, "rb").read() == expect(4, [(bufA, A, 0), (o1, "o1.ctx:2,0", 1)], {3: "z"}, [(3, *r[0])])


# ---- sort / index / table scans ---------------------------------------
def _np_sorted(rec, W, rs):
    a = np.frombuffer(rec, np.uint8).reshape(-1, rs)
    keys = a[:, :8 * W].copy().view("<u8").reshape(-1, W)
    order = np.lexsort([keys[:, w] for w in range(W - 1, -1, -1)])
    return a[order].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("k,ncols", [(31, 1), (63, 2), (33, 3), (5, 1)])
def test_sort_records_and_sorted_check(mcx, k, ncols):
    rng = np.random.default_rng(k + ncols)
    W = (2 * k + 63) // 64
    n = 50000
    keys, covgs, edges, rec = _random_records(rng, k, ncols, n, W)
    # distinct keys (sort has no tie rule): de-duplicate
    rs = 8 * W + 5 * ncols
    a = np.frombuffer(rec, np.uint8).reshape(-1, rs)
    _, first = np.unique(a[:, :8 * W].copy().view("<u8").reshape(-1, W), axis=0, return_index=True)
    rec = a[np.sort(first)].tobytes()
    want = _np_sorted(rec, W, rs)
    got = mcx.sort_records(rec, k, ncols)
    assert got == want
    assert mcx.records_sorted(got, k, ncols) == -1
    nrec = len(rec) // rs
    swapped = bytearray(got)
    i = nrec // 2
    swapped[i * rs:(i + 1) * rs], swapped[(i + 1) * rs:(i + 2) * rs] = got[(i + 1) * rs:(i + 2) * rs], got[i * rs:(i + 1) * rs]
    assert mcx.records_sorted(bytes(swapped), k, ncols) == i + 1
    dup = got[:rs] + got[:rs]
    assert mcx.records_sorted(dup, k, ncols) == 1          # equal keys are "not sorted" (binary_kmer_ge)
    assert mcx.sort_records(b"", k, ncols) == b"" and mcx.sort_records(got[:rs], k, ncols) == got[:rs]


@pytest.mark.gpu
def test_kmer_covg_and_histogram(mcx, orc):
    k, ncols = 31, 3
    g0 = synth.genome(20000, 12)
    og = orc.Graph(k, ncols, 1 << 20)
    g = mcx.Graph(k, ncols, 1 << 20)
    for c in range(ncols - 1):
        b, o = synth.reads(3000 * (c + 1), 100, seed=40 + c, g=g0, n_frac=0.02)
        og.add_reads(c, b, o); g.add_reads(c, b, o)
    body = og.body_bytes(True)
    rs = 8 + 5 * ncols
    a = np.frombuffer(body, np.uint8).reshape(-1, rs)
    cov = a[:, 8:8 + 4 * ncols].copy().view("<u4").reshape(-1, ncols).astype(np.uint64)
    nk, sc = g.kmer_covg()
    assert nk.tolist() == (cov > 0).sum(axis=0).tolist()
    assert sc.tolist() == cov.sum(axis=0).tolist()
    for nbins in (2, 16, 5000):
        want = np.bincount(np.minimum(cov.sum(axis=1), nbins - 1).astype(np.int64), minlength=nbins)
        assert g.covg_histogram(nbins).tolist() == want.tolist()
    g.close()


def _index_expected(buf, block_kmers):
    """ctx_index.c:117-158 restated (with its block arithmetic)"""
    hdr, hs = ctxio.read_header(buf)
    k, W = hdr["kmer_size"], hdr["num_words"]
    km = 8 * W + 5 * hdr["num_cols"]
    bs = block_kmers * km
    lines = ["#block_start\tnext_block\tfirst_kmer\tkmer_idx\tnext_kmer_idx"]
    off, koff, p = hs, 0, hs
    while p + km <= len(buf):
        words = struct.unpack("<%dQ" % W, buf[p:p + 8 * W])
        v = 0
        for w in words:
            v = (v << 64) | w
        kstr = "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))
        bl_bytes = km + min(bs - km, len(buf) - (p + km))
        bl_kmers = 1 + bl_bytes // km
        lines.append("%d\t%d\t%s\t%d\t%d" % (off, off + bl_bytes, kstr, koff, koff + bl_kmers))
        off += bl_bytes; koff += bl_kmers; p += bl_bytes
        if bl_kmers < block_kmers:
            break
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
def test_sort_and_index_commands(mcx, orc, tmp_path):
    k = 31
    g0 = synth.genome(20000, 3)
    r = synth.reads(2000, 100, seed=9, g=g0, n_frac=0.05)
    f = _fasta(tmp_path / "r.fa", *r)
    unsorted = str(tmp_path / "u.ctx")
    rc, _, err = run(31, "build", "-q", "-k", str(k), "-n", "1M", "-s", "a", "--seq", f, "-s", "b", "--seq", f, unsorted)
    assert rc == 0, err
    og, gi = _reads_graph(orc, k, 2, [(0, *r), (1, *r)], ["a", "b"])
    want = ctxio.header_bytes(k, gi) + og.body_bytes(True)
    ubuf = open(unsorted, "rb").read()
    assert ubuf != want and len(ubuf) == len(want)
    # index refuses an unsorted file
    rc, _, err = run(31, "index", unsorted)
    assert rc == 1 and "File is not sorted" in err
    # sort to a new file (header passed through unchanged), refuse to overwrite, then in place
    s1 = str(tmp_path / "s1.ctx")
    rc, _, err = run(31, "sort", "-o", s1, unsorted)
    assert rc == 0, err
    assert open(s1, "rb").read() == want
    rc, _, err = run(31, "sort", "-o", s1, unsorted)
    assert rc == 1 and "File already exists" in err
    rc, _, err = run(31, "sort", "-q", unsorted)
    assert rc == 0 and err == ""
    assert open(unsorted, "rb").read() == want
    # from a stream: -n required
    rc, _, err = run(31, "sort", "-o", str(tmp_path / "s2.ctx"), "-", stdin=ubuf)
    assert rc == 1 and "must give -n" in err
    rc, out, err = run(31, "sort", "-q", "-n", str(og.nkmers), "-o", "-", "-", stdin=ubuf)
    assert rc == 0 and out == want
    rc, _, err = run(31, "sort", "-m", "1K", unsorted)
    assert rc == 1 and "Require at least" in err
    rc, _, err = run(31, "sort", unsorted + ":0")
    assert rc == 1 and "Cannot open graph file with a filter" in err
    # index: default block size (one block) and small blocks
    rc, out, err = run(31, "index", s1)
    assert rc == 0 and out.decode() == _index_expected(want, (4 << 20) // 18)
    idx = str(tmp_path / "s1.idx")
    rc, out, err = run(31, "index", "-b", "100", "-o", idx, s1)
    assert rc == 0 and out == b"" and open(idx).read() == _index_expected(want, 100)
    rc, out, err = run(31, "index", "-s", "1800", s1)
    assert rc == 0 and out.decode() == _index_expected(want, 100)
    rc, _, err = run(31, "index", "-s", "1800", "-b", "100", s1)
    assert rc == 1 and "Cannot use --block-kmers and --block-size together" in err


@pytest.mark.gpu
def test_reference_build0_command_lines(mcx, orc, tmp_path):
    """The reference's own integration test tests/build/build0/Makefile: 60 random bases, k=21,
    `build --sample Wallace --sample Gromit --seq seq.fa --sample Trousers --seq seq.fa --seq2
    seq.fa:seq.fa`, then `sort` in place on a copy and `index` (its later steps -- view, check,
    contigs, rmsubstr -- are other commands).  Expected bytes come from the oracle."""
    k = 21
    rng = np.random.default_rng(60)
    seq = bytes(rng.choice(list(b"ACGT"), 60).astype(np.uint8))
    fa = tmp_path / "seq.fa"
    fa.write_bytes(b">seq\n" + seq + b"\n")
    out = str(tmp_path / "seq.k21.ctx")
    rc, _, err = run(31, "build", "-q", "-m", "1M", "-k", str(k), "--sample", "Wallace", "--sample", "Gromit", "--seq", str(fa),
                     "--sample", "Trousers", "--seq", str(fa), "--seq2", str(fa) + ":" + str(fa), out)
    assert rc == 0, err
    b, o = orc.pack_reads([seq.decode()])
    og, gi = _reads_graph(orc, k, 3, [(1, b, o), (2, b, o), (2, b, o), (2, b, o)], ["Wallace", "Gromit", "Trousers"])
    want_sorted = ctxio.header_bytes(k, gi) + og.body_bytes(True)
    got = open(out, "rb").read()
    hdr, hs = ctxio.read_header(got)
    assert got[:hs] == want_sorted[:hs] and hdr["num_cols"] == 3
    keys, covgs, edges = ctxio.records(got, hdr, hs)
    assert len(keys) == 40 and (covgs[:, 0] == 0).all() and (covgs[:, 1] == 1).all() and (covgs[:, 2] == 3).all()
    assert (edges[:, 0] == 0).all() and (edges[:, 1] == edges[:, 2]).all()
    srt = str(tmp_path / "sort.k21.ctx")
    open(srt, "wb").write(got)
    rc, _, err = run(31, "sort", "-q", srt)
    assert rc == 0, err
    assert open(srt, "rb").read() == want_sorted
    rc, outb, err = run(31, "index", "-q", srt)
    assert rc == 0 and outb.decode() == _index_expected(want_sorted, (4 << 20) // (8 + 15))
    assert len(outb.decode().splitlines()) == 2      # header line + one block


# ---- build --intersect ------------------------------------------------------------------------
def _oracle_intersect(orc, k, ncols, isec_bufs, graph_loads, names, jobs):
    """ctx_build.c:341-413 with the oracle: intersection graphs (flattened), --graph files loaded
    must-exist with the edge mask, must-exist reads, removal of k-mers without coverage and edge
    intersection.  -> expected file bytes"""
    og = orc.Graph(k, ncols, 1 << 20)
    gi = [ctxio.GraphInfo() for _ in range(ncols)]
    for buf in isec_bufs:
        hdr, hs = ctxio.read_header(buf)
        keys, covgs, edges = ctxio.records(buf, hdr, hs)
        for i in range(len(keys)):
            cv = min(0xFFFFFFFF, int(covgs[i].astype(np.uint64).sum()))
            og.add_isec_record(keys[i], cv, int(np.bitwise_or.reduce(edges[i])))
    # (the intersection graphs' headers are merged into colour 0 and then reset: no trace)
    for buf, spec, off in graph_loads:
        hdr, _ = ctxio.read_header(buf)
        _, filt = ctxio.parse_filter(spec, hdr["num_cols"], off)
        ctxio.load_into(og, gi, buf, filt, must_exist=True)
    for c, n in names.items():
        gi[c].sample_name = n
    og.set_must_exist(True)
    for col, b, o in jobs:
        st = og.add_reads(col, b, o)
        gi[col].update_contigs(st.total_bases_loaded, st.contigs_parsed)
    og.isec_finish()
    return ctxio.header_bytes(k, gi) + og.body_bytes(True), og


@pytest.mark.gpu
def test_reference_build1_intersect_and_graph(mcx, orc, tmp_path):
    """The reference's own integration test tests/build/build1/Makefile: two --intersect graphs, a
    --graph loaded into colour 1 and reads into colour 2 must leave exactly 8 k-mers (6 of SEQA_SUB
    via the reads, 2 of SEQB_SUB via the graph; the 100 extra bases are in neither intersection)."""
    K = 11
    SEQA, SEQB = "TCCCTGGTATCAACTTGTCTCTGGTCGGCC", "TACCAATCGGGACAACACGGGTCACTACGA"
    SEQA_SUB, SEQB_SUB = "GGTATCAACTTGTCTC", "CGGGACAACACG"
    EXTRA = "GTTATTAGAATGTTTAATTAATCATAGAAGCAACCTGGGGACACGTCCTCGTGACTGGAGCAGTACAACCCATCAATTAACTCTATTACTATACGGGAAC"

    def fasta(name, seqs):
        p = tmp_path / name
        p.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
        return str(p)

    # dnacat -F -n 9 prints the given sequence and appends 9 random bases: any 9 do
    isec0, isec1 = fasta("isec0.fa", [SEQA + "ACGTTGCAA"]), fasta("isec1.fa", [SEQB + "TTGCAGTCA"])
    seq, small = fasta("seq.fa", [SEQA_SUB, EXTRA]), fasta("small.fa", [SEQB_SUB, EXTRA])
    ctx = {}
    for name, fa in (("isec0", isec0), ("isec1", isec1), ("small", small)):
        ctx[name] = str(tmp_path / (name + ".ctx"))
        rc, _, err = run(31, "build", "-q", "-m", "1M", "-k", str(K), "--sample", name, "--seq", fa, ctx[name])
        assert rc == 0, err
    merged = str(tmp_path / "merge.ctx")
    rc, _, err = run(31, "build", "-q", "-m", "1M", "-k", str(K), "--sort", "--intersect", ctx["isec1"], "--intersect", ctx["isec0"],
                     "--graph", "1:" + ctx["small"], "--sample", "Spiderman", "--seq", seq, merged)
    assert rc == 0, err
    got = open(merged, "rb").read()
    hdr, hs = ctxio.read_header(got)
    keys, covgs, edges = ctxio.records(got, hdr, hs)
    assert hdr["num_cols"] == 3 and len(keys) == 8                      # the reference's assertion
    assert (covgs[:, 0] == 0).all() and sorted((covgs[:, 1] > 0).tolist()) == [False] * 6 + [True] * 2
    assert int((covgs[:, 2] > 0).sum()) == 6
    b, o = orc.pack_reads([SEQA_SUB, EXTRA])
    want, _ = _oracle_intersect(orc, K, 3, [open(ctx["isec1"], "rb").read(), open(ctx["isec0"], "rb").read()],
                                [(open(ctx["small"], "rb").read(), "1:" + ctx["small"], 0)], {2: "Spiderman"}, [(2, b, o)])
    assert got == want
    # ... and the same command with the table split over two and four devices (mcx_multi.h:
    # grp_add_reads_must_exist; consecutive k-mers of a read live on different shards)
    for devs in ("0,0", "0,0,0,0"):
        out = str(tmp_path / ("merge_%d.ctx" % len(devs)))
        rc, _, err = run(31, "build", "-q", "-D", devs, "-n", "1M", "-k", str(K), "--sort", "--intersect", ctx["isec1"], "--intersect", ctx["isec0"],
                         "--graph", "1:" + ctx["small"], "--sample", "Spiderman", "--seq", seq, out)
        assert rc == 0, err
        assert open(out, "rb").read() == want, devs
    rc, _, err = run(31, "build", "-q", "-k", str(K), "-I", ctx["isec0"], "-s", "a", "-p", "--seq", seq, str(tmp_path / "x.ctx"))
    assert rc == 1 and "--remove-pcr" in err


@pytest.mark.gpu
@pytest.mark.parametrize("maxk,k", [(31, 21), (63, 41)])
def test_intersect_matches_oracle_on_random_graphs(mcx, orc, tmp_path, maxk, k):
    """Larger case: a 2-colour intersection graph (flattened) whose reads have errors, reads and a
    --graph that only partly overlap it, lowercase / N-containing reads."""
    g0 = synth.genome(20000, 31)
    g1 = synth.genome(20000, 32)
    ra = synth.reads(1500, 90, seed=1, g=g0, n_frac=0.05)
    rb = synth.reads(800, 90, seed=2, g=g1)
    rq = synth.reads(1200, 100, seed=3, g=np.concatenate([g0[:12000], g1[:8000]]), n_frac=0.05, lower_frac=0.1)
    rg = synth.reads(900, 80, seed=4, g=np.concatenate([g1[:5000], g0[5000:9000]]))
    f = [_fasta(tmp_path / ("r%d.fa" % i), *r) for i, r in enumerate((ra, rb, rq, rg))]
    I = str(tmp_path / "I.ctx"); G = str(tmp_path / "G.ctx"); out = str(tmp_path / "o.ctx")
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "-s", "a", "--seq", f[0], "-s", "b", "--seq", f[1], I)
    assert rc == 0, err
    rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "1M", "-s", "g", "--seq", f[3], G)
    assert rc == 0, err
    rc, _, err = run(maxk, "build", "-k", str(k), "-n", "1M", "--sort", "-I", I, "-g", G, "-s", "q", "--seq", f[2], "--seq", f[0], out)
    assert rc == 0, err
    assert "Flattening intersection graph into colour 0" in err
    want, og = _oracle_intersect(orc, k, 2, [open(I, "rb").read()], [(open(G, "rb").read(), G, 0)], {1: "q"}, [(1, *rq), (1, *ra)])
    got = open(out, "rb").read()
    assert 0 < og.nkmers and got == want
    for devs in ("0,0", "0,0,0,0,0,0,0,0"):  # the table split over several devices: same file
        out2 = str(tmp_path / ("o_%d.ctx" % len(devs)))
        rc, _, err = run(maxk, "build", "-q", "-D", devs, "-k", str(k), "-n", "1M", "--sort", "-I", I, "-g", G, "-s", "q", "--seq", f[2], "--seq", f[0], out2)
        assert rc == 0, err
        assert open(out2, "rb").read() == want, devs


@pytest.mark.gpu
def test_reference_inferedges_build_inputs(mcx, orc, tmp_path):
    """The `build` invocations of the reference's tests/inferedges/Makefile: k=5, plain one-sequence-
    per-line input through process substitution (a pipe: read once, no format probe), an empty input."""
    cases = {
        "RightEdges": "AAGGA\nAAGGC\nAAGGG\nCAAGGT\n",
        "LeftEdges": "ACAAG\nCCAAGG\nGCAAG\nTCAAG",          # no trailing newline
        "MsEmpty": "\n",
    }
    for name, text in cases.items():
        out = str(tmp_path / (name + ".ctx"))
        src = tmp_path / (name + ".txt")
        src.write_text(text)
        cmd = "%s build -q -m 1M -k 5 --sort --sample %s --seq <(cat %s) %s" % (os.path.join(BIN, "mccortex31"), name, src, out)
        p = subprocess.run(["bash", "-c", cmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        reads = [l for l in text.split("\n") if l]
        og, gi = _reads_graph(orc, 5, 1, [(0, *orc.pack_reads(reads))] if reads else [], [name])
        assert open(out, "rb").read() == ctxio.header_bytes(5, gi) + og.body_bytes(True), name
    keys = ctxio.records(*(lambda b: (b,) + ctxio.read_header(b))(open(str(tmp_path / "RightEdges.ctx"), "rb").read()))[0]
    assert len(keys) == 5     # AAGGA, AAGGC, AAGGG, and CAAGG + AAGGT of the 6-base read


@pytest.mark.parametrize("name,maxk", [("tiny_k31", 31), ("tiny_k63", 63), ("tiny_k5", 31)])
def test_index_command_on_golden_files_cpu(mcx, name, maxk, tmp_path):
    """`index` needs no GPU (its device sortedness check is skipped without one): the table it
    prints for the committed golden graphs equals the restatement of ctx_index.c:117-158."""
    path = os.path.join(GOLD, name + ".ctx")
    buf = open(path, "rb").read()
    hdr, hs = ctxio.read_header(buf)
    km = 8 * hdr["num_words"] + 5 * hdr["num_cols"]
    rc, out, err = run(maxk, "index", "-q", path)
    assert rc == 0, err
    assert out.decode() == _index_expected(buf, (4 << 20) // km)
    idx = str(tmp_path / "g.idx")
    rc, out, err = run(maxk, "index", "-b", "7", "-o", idx, path)
    assert rc == 0 and out == b"" and "[index] block bytes: %d kmers: 7" % (7 * km) in err
    assert open(idx).read() == _index_expected(buf, 7)
    # an unsorted file is refused by the host-side check of consecutive block starts
    n = (len(buf) - hs) // km
    if n >= 16:
        recs = [buf[hs + i * km: hs + (i + 1) * km] for i in range(n)]
        recs[0], recs[8] = recs[8], recs[0]
        bad = tmp_path / "bad.ctx"
        bad.write_bytes(buf[:hs] + b"".join(recs))
        rc, _, err = run(maxk, "index", "-b", "4", str(bad))
        assert rc == 1 and "File is not sorted" in err
