"""`mccortex<K> build` host program: command-line contract (CPU) and end-to-end .ctx parity (GPU).
The option rules follow src/commands/ctx_build.c:133-242; the end-to-end check is the reference's
own integration pattern (tests/sort/Makefile:29-45): `build --sort` output compared byte for byte."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mccortex_amd", "bin")


def run(maxk, *args, stdin=None):
    exe = os.path.join(BIN, "mccortex%d" % maxk)
    p = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, input=stdin)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


@pytest.fixture(scope="module")
def built(mcx):
    assert os.path.exists(os.path.join(BIN, "mccortex31")) and os.path.exists(os.path.join(BIN, "mccortex63"))
    return True


def test_usage_and_argument_errors(built, tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_text(">r\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    rc, _, err = run(31)
    assert rc == 1 and "usage: mccortex31 <command>" in err
    rc, _, err = run(31, "view", "x.ctx")
    assert rc == 1 and "not part of this build" in err
    rc, _, err = run(31, "build", "-h")
    assert rc == 1 and "usage: mccortex31 build [options] <out.ctx>" in err
    cases = [
        (["build", "-k", "30", "-s", "a", "--seq", str(fa), "o.ctx"], "Invalid kmer-size (30)"),
        (["build", "-k", "33", "-s", "a", "--seq", str(fa), "o.ctx"], "Please recompile with correct kmer size (33)"),
        (["build", "-k", "31", "--seq", str(fa), "o.ctx"], "Please give sample name first"),
        (["build", "-k", "31", "-s", "a", "--seq", str(fa)], "Expected exactly one graph file"),
        (["build", "-k", "31", "-s", "a", "--seq", str(fa), "o.ctx", "extra"], "Expected only one graph file"),
        (["build", "-k", "31", "o.ctx"], "No inputs given"),
        (["build", "-k", "31", "-s", "a", "--seq", str(fa), "-Q", "10", "o.ctx"], "Arguments not given BEFORE sequence file"),
        (["build", "-s", "a", "--seq", str(fa), "o.ctx"], "kmer size not set"),
        (["build", "-k", "31", "-k", "21", "-s", "a", "--seq", str(fa), "o.ctx"], "given twice"),
        (["build", "-k", "31", "-s", "undefined", "--seq", str(fa), "o.ctx"], "Bad sample name"),
        (["build", "-k", "31", "-s", "a b", "--seq", str(fa), "o.ctx"], "whitespace"),
        (["build", "-k", "31", "-s", "a", "--seq", str(tmp_path / "missing.fa"), "o.ctx"], "Cannot open"),
        (["build", "-k", "31", "-m", "12XB", "-s", "a", "--seq", str(fa), "o.ctx"], "Invalid memory argument"),
        (["build", "-k", "31", "-s", "a", "-p", "--seq", str(fa), "o.ctx"], "remove PCR duplicates: yes"),
        (["build", "-k", "31", "-s", "a", "-p", "--seq2", str(fa) + ":" + str(tmp_path / "missing.fa"), "o.ctx"], "Cannot open -2 file"),
        (["build", "-k", "31", "-s", "a", "-M", "XX", "--seq", str(fa), "o.ctx"], "must be one of: FF,FR,RF,RR"),
        (["build", "-k", "31", "-s", "a", "--bogus", "o.ctx"], "Bad option"),
    ]
    for args, msg in cases:
        rc, _, err = run(31, *args)
        assert rc == 1, args
        assert msg in err, (args, err)
    # mccortex63 accepts 33..63 only (MIN_KMER_SIZE = MAXK-30, reference Makefile:48)
    rc, _, err = run(63, "build", "-k", "31", "-s", "a", "--seq", str(fa), "o.ctx")
    assert rc == 1 and "Please recompile" in err
    # single-dash long options are accepted (getopt_long_only)
    rc, _, err = run(31, "build", "-kmer", "30", "-sample", "a", "-seq", str(fa), "o.ctx")
    assert rc == 1 and "Invalid kmer-size (30)" in err


def test_hashtest_usage_and_argument_errors(built):
    """`hashtest` (src/commands/ctx_exp_hashtest.c:78-122): option rules and messages, no device needed."""
    rc, _, err = run(31, "hashtest")
    assert rc == 1 and "usage: mccortex31 hashtest [options] <num_ops>" in err and "31 >= k >= 3" in err
    for args, msg in [(["hashtest", "1000"], "kmer size not set with -k <K>"),
                      (["hashtest", "-k", "33", "1000"], "Please recompile with correct kmer size (33)"),
                      (["hashtest", "-k", "30", "1000"], "Invalid kmer-size (30): requires odd number 3 <= k <= 31"),
                      (["hashtest", "-k", "31", "-k", "21", "1000"], "given twice"),
                      (["hashtest", "-k", "31"], "usage: mccortex31 hashtest"),
                      (["hashtest", "-k", "31", "12x"], "Invalid <num_ops>"),
                      (["hashtest", "-k", "31", "-m", "12XB", "10"], "Invalid memory argument"),
                      (["hashtest", "-k", "31", "--bogus", "10"], "Bad option")]:
        rc, _, err = run(31, *args)
        assert rc == 1 and msg in err, (args, err)
    rc, _, err = run(63, "hashtest", "-k", "31", "10")
    assert rc == 1 and "Please recompile" in err


@pytest.mark.gpu
def test_hashtest_command(built, orc):
    """Integer keys 0..N-1 into the table on the device: every key is new, so `filled` must be N; the -F mode must
    print the reference's sum-of-XORs of bklk3 hashes for the same -t (oracle: orc_kmer_hash, pinned against the
    reference's own kmer_hash.h / lookup3.h in tests/test_oracle.py)."""
    n = 200_000
    rc, _, err = run(31, "hashtest", "-k", "31", "-n", "1M", str(n))
    assert rc == 0, err
    # (the table in HBM: the 2^20 hash-addressed slots -n asks for + the overflow area, 1 / 32 of them)
    assert "filled: 0 / 1,081,344 (0.00%)" in err and "filled: 200,000 / 1,081,344 (18.50%)" in err
    assert "using 1 thread (single-threaded code)" in err and "Output hash: 0" in err
    rc, _, err = run(63, "hashtest", "-k", "63", "-n", "1M", "-t", "4", str(n))
    assert rc == 0 and "filled: 200,000 / 1,081,344" in err and "using 4 threads (multi-threaded code)" in err, err
    # more keys than the table holds: the reference's message
    rc, _, err = run(31, "hashtest", "-k", "31", "-n", "64K", str(n))
    assert rc == 1 and "Hash table is full" in err, err
    for maxk, k, t in ((31, 31, 1), (31, 21, 3), (63, 63, 4)):
        want = orc.lib().orc_hashtest_func(k, n, t)
        rc, _, err = run(maxk, "hashtest", "-k", str(k), "-F", "-t", str(t), str(n))
        assert rc == 0 and ("Output hash: %d" % want) in err, (k, t, want, err[-300:])


def _write_inputs(tmp_path, bases, offs, name, fmt, gz=False, qual=None, width=0):
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]
    out = []
    for i, r in enumerate(reads):
        if fmt == "fa":
            body = r if not width else b"\n".join(r[j:j + width] for j in range(0, max(len(r), 1), width))
            out.append(b">read%d desc\n" % i + body + b"\n")
        elif fmt == "fq":
            q = bytes(qual[int(offs[i]):int(offs[i + 1])]) if qual is not None else b"I" * len(r)
            out.append(b"@read%d\n" % i + r + b"\n+\n" + q + b"\n")
        else:
            if len(r):
                out.append(r + b"\n")
    data = b"".join(out)
    p = tmp_path / (name + "." + fmt + (".gz" if gz else ""))
    if gz:
        with gzip.open(p, "wb") as f:
            f.write(data)
    else:
        p.write_bytes(data)
    return str(p)


@pytest.mark.gpu
def test_build_sort_matches_oracle_ctx(built, orc, tmp_path):
    g = synth.genome(40000, 3)
    b0, o0 = synth.reads(3000, 100, seed=1, g=g, n_frac=0.05, lower_frac=0.1)
    b1, o1 = synth.reads(2000, 150, seed=2, g=g, n_frac=0.05)
    b2, o2 = synth.reads(1500, 80, seed=3, g=g, var_len=True)
    f0 = _write_inputs(tmp_path, b0, o0, "s0", "fa", width=60)
    f1 = _write_inputs(tmp_path, b1, o1, "s1", "fq", gz=True)
    f2 = _write_inputs(tmp_path, b2, o2, "s2", "txt")
    for maxk, k in [(31, 31), (31, 17), (63, 51)]:
        out = str(tmp_path / ("out%d.ctx" % k))
        rc, _, err = run(maxk, "build", "-k", str(k), "-n", "1M", "--sort",
                         "--sample", "alice", "--seq", f0, "--seq", f1, "--sample", "bob", "--seq2", f2 + ":" + f0, out)
        assert rc == 0, err
        og = orc.Graph(k, 2, 1 << 20)
        og.set_sample(0, "alice"); og.set_sample(1, "bob")
        # plain-format empty lines are skipped by the reader: drop empty reads for colour 1 file 1
        keep = np.diff(o2.astype(np.int64)) > 0
        o2k = np.concatenate([[0], np.cumsum(np.diff(o2.astype(np.int64))[keep])]).astype(np.uint64)
        for col, bb, oo in [(0, b0, o0), (0, b1, o1), (1, b2, o2k), (1, b0, o0)]:
            st = og.add_reads(col, bb, oo)
            og.update_stats(col, st)
        want = og.ctx_bytes(True)
        got = open(out, "rb").read()
        assert len(got) == len(want)
        assert got == want
        # refuses to overwrite without -f; unsorted output has the same record set
        rc, _, err = run(maxk, "build", "-k", str(k), "-n", "1M", "-s", "alice", "--seq", f0, out)
        assert rc == 1 and "already exists" in err
        rc, _, err = run(maxk, "build", "-q", "-f", "-k", str(k), "-n", "1M", "-s", "alice", "--seq", f0, "--seq", f1,
                         "-s", "bob", "--seq", f2, "--seq", f0, out)
        assert rc == 0 and err == ""
        got2 = open(out, "rb").read()
        hs = og.header_size()
        rs = 8 * og.W + 10
        assert got2[:hs] == want[:hs]
        a = np.frombuffer(got2[hs:], np.uint8).reshape(-1, rs)
        b = np.frombuffer(want[hs:], np.uint8).reshape(-1, rs)
        assert sorted(map(bytes, a)) == sorted(map(bytes, b))


@pytest.mark.gpu
def test_build_quality_cutoff_stdin_stdout(built, orc, tmp_path):
    bases, offs = synth.reads(2000, 100, genome_len=20000, seed=8, n_frac=0.05)
    rng = np.random.default_rng(1)
    quals = rng.integers(33, 74, len(bases)).astype(np.uint8)
    fq = _write_inputs(tmp_path, bases, offs, "q", "fq", qual=quals)
    out = str(tmp_path / "q.ctx")
    rc, _, err = run(31, "build", "-k", "21", "-n", "1M", "-S", "-s", "smp", "-Q", "10", "-O", "33", "-H", "6", "--seq", fq, out)
    assert rc == 0, err
    og = orc.Graph(21, 1, 1 << 20)
    og.set_sample(0, "smp")
    st = og.add_reads(0, bases, offs, quals=quals, fq_cutoff=43, hp_cutoff=6)
    og.update_stats(0, st)
    assert open(out, "rb").read() == og.ctx_bytes(True)
    # stdin '-' as input, '-' (stdout) as output
    fa = open(_write_inputs(tmp_path, bases, offs, "p", "fa"), "rb").read()
    rc, stdout, err = run(31, "build", "-q", "-k", "21", "-n", "1M", "-S", "-s", "smp", "--seq", "-", "-", stdin=fa)
    assert rc == 0, err
    og2 = orc.Graph(21, 1, 1 << 20)
    og2.set_sample(0, "smp")
    st = og2.add_reads(0, bases, offs)
    og2.update_stats(0, st)
    assert stdout == og2.ctx_bytes(True)


@pytest.mark.gpu
def test_build_table_full_dies(built, tmp_path):
    bases, offs = synth.reads(4000, 100, genome_len=300000, seed=4)
    fa = _write_inputs(tmp_path, bases, offs, "big", "fa")
    rc, _, err = run(31, "build", "-k", "31", "-n", "1024", "-s", "a", "--seq", fa, str(tmp_path / "o.ctx"))
    assert rc == 1 and "Hash table is full" in err


@pytest.mark.gpu
def test_chunk_seams_and_long_sequences(built, orc, tmp_path):
    """Reads longer than a staging chunk (chromosome-like FASTA records) and reads that straddle
    chunk boundaries: MCX_STAGE_BYTES shrinks the 32 MiB staging chunk so the 128-byte carry and
    the [pos_lo,pos_hi) ownership of k-mer start positions are exercised thousands of times."""
    g = synth.genome(300_000, 77)
    g[1000:1040] = ord("N")
    g[150_000] = ord("n")
    long_reads = [bytes(g[:120_000]), bytes(g[100_000:300_000]), b"ACGT" * 3000]
    b1, o1 = orc.pack_reads(long_reads)
    b2, o2 = synth.reads(3000, 151, seed=5, g=g, n_frac=0.1)
    f1 = _write_inputs(tmp_path, b1, o1, "long", "fa", width=70)
    f2 = _write_inputs(tmp_path, b2, o2, "short", "fq")
    for stage in ("1024", "4096", "65536"):
        for maxk, k in [(31, 31), (63, 63)]:
            out = str(tmp_path / ("seam_%s_%d.ctx" % (stage, k)))
            exe = os.path.join(BIN, "mccortex%d" % maxk)
            env = dict(os.environ, MCX_STAGE_BYTES=stage)
            p = subprocess.run([exe, "build", "-q", "-f", "-k", str(k), "-n", "2M", "-S", "-s", "smp", "--seq", f1, "--seq", f2, out],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()
            og = orc.Graph(k, 1, 1 << 21)
            og.set_sample(0, "smp")
            for bb, oo in [(b1, o1), (b2, o2)]:
                st = og.add_reads(0, bb, oo)
                og.update_stats(0, st)
            assert open(out, "rb").read() == og.ctx_bytes(True), (stage, k)


@pytest.mark.gpu
def test_parallel_ingest_matches_sequential_and_oracle(built, orc, tmp_path):
    """-t N parses uncompressed files > 1 MB with N threads over byte ranges (host/par_ingest.c):
    FASTQ (incl. qualities that start with '@' and '+'), wrapped FASTA, plain; -Q through the fast
    path; a multi-line FASTQ is declined by the probe and parsed sequentially."""
    g = synth.genome(400_000, 11)
    b, o = synth.reads(30000, 150, seed=3, g=g, n_frac=0.05, lower_frac=0.02)
    rng = np.random.default_rng(9)
    quals = rng.integers(33, 75, len(b)).astype(np.uint8)
    quals[o[:-1].astype(np.int64)[::3]] = ord("@")   # quality lines that look like headers
    quals[o[:-1].astype(np.int64)[1::3]] = ord("+")
    fq = _write_inputs(tmp_path, b, o, "par", "fq", qual=quals)
    fa = _write_inputs(tmp_path, b, o, "par", "fa", width=61)
    tx = _write_inputs(tmp_path, b, o, "par", "txt")
    assert os.path.getsize(fq) > (1 << 20) and os.path.getsize(fa) > (1 << 20)

    def oracle(**kw):
        og = orc.Graph(31, 1, 1 << 22)
        og.set_sample(0, "s")
        st = og.add_reads(0, b, o, **kw)
        og.update_stats(0, st)
        return og.ctx_bytes(True)

    want = oracle()
    for f in (fq, fa, tx):
        for t in ("1", "3", "8"):
            out = str(tmp_path / "p.ctx")
            rc, _, err = run(31, "build", "-f", "-t", t, "-k", "31", "-n", "4M", "-S", "-s", "s", "--seq", f, out)
            assert rc == 0, err
            assert open(out, "rb").read() == want, (f, t)
            assert "SE reads: 30,000" in err and "bases read: 4,500,000" in err
    wantq = oracle(quals=quals, fq_cutoff=33 + 15, hp_cutoff=0)
    for t in ("1", "5"):
        out = str(tmp_path / "q.ctx")
        rc, _, err = run(31, "build", "-f", "-t", t, "-k", "31", "-n", "4M", "-S", "-s", "s", "-Q", "15", "--seq", fq, out)
        assert rc == 0, err
        assert open(out, "rb").read() == wantq, t
    # multi-line FASTQ: sequence and quality wrapped at 50 columns
    recs = []
    for i in range(len(o) - 1):
        r = bytes(b[int(o[i]):int(o[i + 1])])
        q = b"I" * len(r)
        recs.append(b"@r%d\n" % i + b"\n".join(r[j:j + 50] for j in range(0, len(r), 50)) + b"\n+\n" +
                    b"\n".join(q[j:j + 50] for j in range(0, len(q), 50)) + b"\n")
    ml = tmp_path / "multi.fq"
    ml.write_bytes(b"".join(recs))
    out = str(tmp_path / "m.ctx")
    rc, _, err = run(31, "build", "-f", "-t", "4", "-k", "31", "-n", "4M", "-S", "-s", "s", "--seq", str(ml), out)
    assert rc == 0, err
    assert open(out, "rb").read() == want
    # a regular head (4-line records beyond the 4 MB that are probed), wrapped records after it: the
    # range parsers give up after batches were submitted; the graph is emptied and the file read again
    head = []
    for i in range(len(o) - 1):
        r = bytes(b[int(o[i]):int(o[i + 1])])
        head.append(b"@h%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    late = tmp_path / "late.fq"
    late.write_bytes(b"".join(head) + b"".join(recs))
    assert sum(map(len, head)) > (5 << 20)
    og = orc.Graph(31, 1, 1 << 22)
    og.set_sample(0, "s")
    for _ in range(2):
        og.update_stats(0, og.add_reads(0, b, o))
    rc, _, err = run(31, "build", "-f", "-t", "4", "-k", "31", "-n", "4M", "-S", "-s", "s", "--seq", str(late), out)
    assert rc == 0, err
    assert "reading the file again with one parser thread" in err
    assert open(out, "rb").read() == og.ctx_bytes(True)
    assert "SE reads: 60,000" in err


@pytest.mark.gpu
def test_build_remove_pcr(built, orc, tmp_path):
    """--remove-pcr with single, paired (--seq2) and interleaved (--seqi) inputs, --keep-pcr in between,
    a second colour (read starts wiped, ctx_build.c:389-395): whole file against the restatement"""
    k = 31
    g = synth.genome(800, 5)
    b1, o1 = synth.reads(3000, 100, seed=1, g=g, n_frac=0.05)          # mates 1
    b2, o2 = synth.reads(3000, 100, seed=2, g=g, n_frac=0.05)          # mates 2
    b3, o3 = synth.reads(2000, 80, seed=3, g=g, lower_frac=0.2)        # single reads
    rng = np.random.default_rng(4)
    q1 = rng.integers(40, 74, len(b1)).astype(np.uint8)
    q2 = rng.integers(40, 74, len(b2)).astype(np.uint8)
    f1 = _write_inputs(tmp_path, b1, o1, "m1", "fq", qual=q1)
    f2 = _write_inputs(tmp_path, b2, o2, "m2", "fq", qual=q2, gz=True)
    f3 = _write_inputs(tmp_path, b3, o3, "se", "fa")
    # interleaved file: m1[0], m2[0], m1[1], ...
    n = len(o1) - 1
    reads = []
    for i in range(n):
        reads.append(bytes(b1[int(o1[i]):int(o1[i + 1])]))
        reads.append(bytes(b2[int(o2[i]):int(o2[i + 1])]))
    bi, oi = orc.pack_reads(reads)
    fi = _write_inputs(tmp_path, bi, oi, "il", "fa")
    out = str(tmp_path / "pcr.ctx")
    rc, _, err = run(31, "build", "-k", str(k), "-n", "64K", "--sort", "-Q", "10", "-O", "33", "-H", "7",
                     "--sample", "a", "--remove-pcr", "--seq", f3, "--matepair", "FR", "--seq2", f1 + ":" + f2,
                     "--keep-pcr", "--seq", f3, "--sample", "b", "--remove-pcr", "-M", "RF", "--seqi", fi, "--seq", f3, out)
    assert rc == 0, err
    assert "remove PCR duplicates: yes" in err and "dup SE reads:" in err
    og = orc.Graph(k, 2, 1 << 16)
    og.set_sample(0, "a"); og.set_sample(1, "b")
    # colour 0: single reads (FASTA: no qualities), then the pairs with qualities, then the same single reads unfiltered
    st, d0 = og.add_reads_pcr(0, b3, o3, hp_cutoff=7, matedir="FR")
    og.update_stats(0, st)
    bp, op = orc.pack_reads([r for i in range(n) for r in (bytes(b1[int(o1[i]):int(o1[i + 1])]), bytes(b2[int(o2[i]):int(o2[i + 1])]))])
    qp = np.frombuffer(b"".join(bytes(x) for i in range(n) for x in (q1[int(o1[i]):int(o1[i + 1])], q2[int(o2[i]):int(o2[i + 1])])), np.uint8)
    st, d1 = og.add_reads_pcr(0, bp, op, quals=qp, fq_cutoff=43, hp_cutoff=7, paired=True, matedir="FR")
    og.update_stats(0, st)
    st = og.add_reads(0, b3, o3, hp_cutoff=7)
    og.update_stats(0, st)
    # colour 1: read starts forgotten
    og.pcr_reset()
    st, d2 = og.add_reads_pcr(1, bi, oi, hp_cutoff=7, paired=True, matedir="RF")
    og.update_stats(1, st)
    st, d3 = og.add_reads_pcr(1, b3, o3, hp_cutoff=7, matedir="RF")
    og.update_stats(1, st)
    assert d0[0] > 0 and d1[1] > 0 and d2[1] > 0 and d3[0] > 0
    for d in (d0[0], d3[0]):
        assert "dup SE reads: %s  dup PE pairs: 0" % format(d, ",") in err
    for d in (d1[1], d2[1]):
        assert "dup SE reads: 0  dup PE pairs: %s" % format(d, ",") in err
    assert open(out, "rb").read() == og.ctx_bytes(True)
    # mate files of different length are an error
    short = _write_inputs(tmp_path, b1[:int(o1[10])], o1[:11], "short", "fa")
    rc, _, err = run(31, "build", "-f", "-k", str(k), "-n", "64K", "-s", "a", "-p", "--seq2", f1 + ":" + short, out)
    assert rc == 1 and "Different number of reads" in err


@pytest.mark.gpu
def test_gzip_inputs_are_read_ahead(built, orc, tmp_path):
    """several .gz files (inflated ahead by reader threads, -t 4) mixed with plain ones, two colours, -Q:
    same file as the restatement; -t 1 (no read-ahead) gives the same bytes"""
    g = synth.genome(30000, 9)
    sets = []
    for i in range(5):
        b, o = synth.reads(1500 + 300 * i, 90 + 10 * i, seed=20 + i, g=g, n_frac=0.05, lower_frac=0.1)
        q = np.random.default_rng(i).integers(35, 74, len(b)).astype(np.uint8)
        fmt, gz = [("fq", True), ("fa", True), ("fq", False), ("fq", True), ("fa", True)][i]
        sets.append((b, o, q if fmt == "fq" else None, _write_inputs(tmp_path, b, o, "in%d" % i, fmt, gz=gz, qual=q if fmt == "fq" else None)))
    empty = tmp_path / "empty.fa.gz"
    with gzip.open(empty, "wb") as f:
        f.write(b"")
    out = str(tmp_path / "ra.ctx")
    args = ["build", "-k", "31", "-n", "1M", "--sort", "-Q", "12", "-O", "33", "-s", "a", "--seq", sets[0][3], "--seq", sets[1][3],
            "--seq", sets[2][3], "--seq", str(empty), "-s", "b", "--seq", sets[3][3], "--seq", sets[4][3]]
    rc, _, err = run(31, *(args[:1] + ["-t", "4"] + args[1:] + [out]))
    assert rc == 0, err
    og = orc.Graph(31, 2, 1 << 20)
    og.set_sample(0, "a"); og.set_sample(1, "b")
    for col, (b, o, q, _) in zip((0, 0, 0, 1, 1), sets):
        st = og.add_reads(col, b, o, quals=q, fq_cutoff=45 if q is not None else 0)
        og.update_stats(col, st)
        if col == 0 and q is not None and _ == sets[2][3]:
            og.update_stats(0, orc.Stats())     # the empty file: a file with no reads still updates the colour's statistics
    want = og.ctx_bytes(True)
    got = open(out, "rb").read()
    assert got == want
    out1 = str(tmp_path / "ra1.ctx")
    rc, _, err = run(31, *(args[:1] + ["-t", "1"] + args[1:] + [out1]))
    assert rc == 0, err
    assert open(out1, "rb").read() == want


@pytest.mark.gpu
def test_build_on_several_devices_matches_oracle_ctx(built, orc, tmp_path):
    """`build -D 0,0[,0,0]`: one table split over several devices behind the same command line
    (csrc/mcx_multi.h; the test box has one GPU, so the same device is named 2 / 4 times): two
    samples, gzip'd and plain inputs, -Q / -H, --remove-pcr pairs, a --graph load, sorted and
    unsorted output -- byte-identical to the oracle's .ctx (ctx_build.c:384-407 is what is replaced)."""
    g = synth.genome(30000, 5)
    b0, o0 = synth.reads(4000, 100, seed=1, g=g, n_frac=0.05, lower_frac=0.1)
    b1, o1 = synth.reads(3000, 120, seed=2, g=g, n_frac=0.05)
    rng = np.random.default_rng(2)
    q1 = rng.integers(33, 74, len(b1)).astype(np.uint8)
    f0 = _write_inputs(tmp_path, b0, o0, "m0", "fa", width=60)
    f1 = _write_inputs(tmp_path, b1, o1, "m1", "fq", gz=True, qual=q1)
    for devs, (maxk, k) in [("0,0", (31, 31)), ("0,0,0,0", (31, 21)), ("0,0", (63, 41))]:
        out = str(tmp_path / ("multi%d.ctx" % k))
        rc, _, err = run(maxk, "build", "-D", devs, "-k", str(k), "-n", "4M", "--sort", "-t", "3",
                         "--sample", "alice", "--seq", f0, "--sample", "bob", "-Q", "10", "-O", "33", "-H", "7", "--seq", f1, out)
        assert rc == 0, err
        assert "device 0" in err
        og = orc.Graph(k, 2, 1 << 22)
        og.set_sample(0, "alice"); og.set_sample(1, "bob")
        st = og.add_reads(0, b0, o0); og.update_stats(0, st)
        st = og.add_reads(1, b1, o1, quals=q1, fq_cutoff=43, hp_cutoff=7); og.update_stats(1, st)
        want = og.ctx_bytes(True)
        assert open(out, "rb").read() == want, devs
        # the graph just written, loaded back with --graph into a 1-colour build next to new reads
        out2 = str(tmp_path / ("multi%d_b.ctx" % k))
        rc, _, err = run(maxk, "build", "-q", "-D", devs, "-k", str(k), "-n", "4M", "--sort", "--graph", "0:" + out + ":0",
                         "--sample", "carol", "--seq", f0, out2)
        assert rc == 0, err
        rc, _, err = run(maxk, "build", "-q", "-k", str(k), "-n", "4M", "--sort", "--graph", "0:" + out + ":0",
                         "--sample", "carol", "--seq", f0, out2 + ".one")
        assert rc == 0, err
        assert open(out2, "rb").read() == open(out2 + ".one", "rb").read()
    # --remove-pcr pairs on two devices == on one
    p1 = _write_inputs(tmp_path, b0, o0, "p1", "fq")
    b2, o2 = synth.reads(4000, 100, seed=9, g=g)
    p2 = _write_inputs(tmp_path, b2, o2, "p2", "fq")
    for kk in ("27", "31"):  # k = 27: exchange format v2 (hash-prefix owners); k = 31: v3 (minimizer owners)
        outs = []
        for devs in ("0", "0,0", "0,0,0,0"):
            out = str(tmp_path / ("pcr_%s_%d.ctx" % (kk, len(devs))))
            rc, _, err = run(31, "build", "-q", "-D", devs, "-k", kk, "-n", "4M", "--sort", "-s", "x", "--remove-pcr", "--seq2", p1 + ":" + p2, out)
            assert rc == 0, err
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000, kk
    # an odd number of devices is refused
    rc, _, err = run(31, "build", "-D", "0,0,0", "-k", "31", "-s", "a", "--seq", f0, str(tmp_path / "x.ctx"))
    assert rc == 1 and "power of two" in err


@pytest.mark.gpu
def test_quality_offset_is_decided_once_per_file(built, orc, tmp_path):
    """Uniformly high Phred+33 qualities ('I' everywhere with a few low ones late in the file) are
    Sanger: -Q 10 must cut at 43, whatever -t is and whichever byte range is parsed first."""
    bases, offs = synth.reads(30000, 100, genome_len=50000, seed=31)
    quals = np.full(len(bases), ord("I"), dtype=np.uint8)
    quals[len(quals) // 2::97] = ord("#")  # low qualities only in the second half of the file
    fq = _write_inputs(tmp_path, bases, offs, "hi", "fq", qual=quals)
    og = orc.Graph(21, 1, 1 << 22)
    og.set_sample(0, "s")
    st = og.add_reads(0, bases, offs, quals=quals, fq_cutoff=43)
    og.update_stats(0, st)
    want = og.ctx_bytes(True)
    for t in ("1", "2", "7"):
        out = str(tmp_path / ("hi%s.ctx" % t))
        rc, _, err = run(31, "build", "-q", "-t", t, "-k", "21", "-n", "4M", "-S", "-s", "s", "-Q", "10", "--seq", fq, out)
        assert rc == 0, err
        assert open(out, "rb").read() == want, t


@pytest.mark.gpu
def test_quality_cutoff_on_a_fifo_keeps_the_head_of_the_stream(built, orc, tmp_path):
    """`-Q` without --fq-offset on an input that can only be read once (a FIFO, <(zcat x.fq.gz)): the
    offset probe must not open it -- it would swallow the first records -- and the offset is latched
    from the first batch instead.  Same .ctx as from the regular file."""
    import threading
    bases, offs = synth.reads(20000, 100, genome_len=40000, seed=77)
    quals = np.full(len(bases), ord("I"), dtype=np.uint8)
    quals[5::53] = ord("#")
    fq = _write_inputs(tmp_path, bases, offs, "ff", "fq", qual=quals)
    og = orc.Graph(21, 1, 1 << 22)
    og.set_sample(0, "s")
    st = og.add_reads(0, bases, offs, quals=quals, fq_cutoff=43)
    og.update_stats(0, st)
    want = og.ctx_bytes(True)
    fifo = str(tmp_path / "reads.fifo")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(open(fq, "rb").read())
    th = threading.Thread(target=feed)
    th.start()
    out = str(tmp_path / "fifo.ctx")
    rc, _, err = run(31, "build", "-q", "-t", "2", "-k", "21", "-n", "4M", "-S", "-s", "s", "-Q", "10", "--seq", fifo, out)
    th.join()
    assert rc == 0, err
    assert open(out, "rb").read() == want


@pytest.mark.gpu
def test_c5_shape_four_samples_on_eight_devices(built, orc, tmp_path):
    """Config C5 through the command line: four samples (four colours, per-colour coverage and edges)
    on a table split over EIGHT devices (`-D 0,0,0,0,0,0,0,0`: the box has one GPU, every shard is on it;
    the exchange, the owners' splits and the per-colour flushes are the real path), inputs of the
    samples alternating as a population build lists them -- byte-identical to the oracle's sorted .ctx
    and to the same build on one device."""
    g = synth.genome(60000, 55)
    files, jobs = [], []
    for smp in range(4):
        for part in range(2):
            b, o = synth.reads(2500, 150, seed=500 + 10 * smp + part, g=g, n_frac=0.02, err=0.002)
            files.append((smp, _write_inputs(tmp_path, b, o, "c5_%d_%d" % (smp, part), "fq", gz=bool(part))))
            jobs.append((smp, b, o))
    og = orc.Graph(31, 4, 1 << 22)
    args = []
    for smp in range(4):
        og.set_sample(smp, "s%d" % smp)
        args += ["--sample", "s%d" % smp]
        for c, f in files:
            if c == smp:
                args += ["--seq", f]
    for smp, b, o in jobs:
        st = og.add_reads(smp, b, o)
        og.update_stats(smp, st)
    want = og.ctx_bytes(True)
    outs = {}
    for devs in ("0,0,0,0,0,0,0,0", "0"):
        out = str(tmp_path / ("c5_%d.ctx" % len(devs)))
        rc, _, err = run(31, "build", "-q", "-D", devs, "-k", "31", "-n", "8M", "-t", "4", "--sort", *args, out)
        assert rc == 0, err
        outs[devs] = open(out, "rb").read()
    assert outs["0"] == want
    assert outs["0,0,0,0,0,0,0,0"] == want
