"""Host plumbing of the multi-GPU build tool: byte ranges of sequence files -> separator streams."""
import re

import numpy as np
import pytest
import torch

import synth
from mccortex_amd import seqstream


def _reads(n=400, seed=1):
    bases, offs = synth.reads(n, 60, genome_len=3000, seed=seed, n_frac=0.1, lower_frac=0.1, var_len=True)
    return [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(n)]


def _contigs(data):
    return sorted(re.findall(rb"[ACGTacgt]+", data))


def _write(tmp_path, reads, fmt, width=0):
    out = []
    for i, r in enumerate(reads):
        if fmt == "fasta":
            body = r if not width else b"\n".join(r[j:j + width] for j in range(0, max(len(r), 1), width))
            out.append(b">r%d some text ACGT\n" % i + body + b"\n")
        elif fmt == "fastq":
            out.append(b"@r%d\n" % i + r + b"\n+\n" + b"@" * len(r) + b"\n")   # '@' qualities: the hard case for alignment
        else:
            out.append(r + b"\n")
    p = tmp_path / ("in." + fmt)
    p.write_bytes(b"".join(out))
    return str(p)


@pytest.mark.parametrize("fmt,width", [("fasta", 0), ("fasta", 17), ("fastq", 0), ("plain", 0)])
@pytest.mark.parametrize("world,step", [(1, 1 << 20), (2, 3000), (3, 700), (5, 1 << 20)])
def test_ranges_cover_every_read_once(tmp_path, fmt, width, world, step):
    reads = _reads()
    path = _write(tmp_path, reads, fmt, width)
    assert seqstream.detect_format(path) == fmt
    got = []
    prev_end = 0
    for rank in range(world):
        for a, b in seqstream.plan_steps(path, fmt, rank, world, step):
            assert a == prev_end and b > a          # the steps tile the file in order, no gap, no overlap
            prev_end = b
            s = seqstream.to_stream(seqstream.load_bytes(path, a, b, "cpu"), fmt)
            got.append(bytes(s.numpy()))
    import os
    assert prev_end == os.path.getsize(path)
    # joining steps with a separator must give the reads' contigs: nothing lost, nothing glued together
    assert _contigs(b"\n".join(got)) == _contigs(b"\n".join(reads))
    # and inside a stream every read is one piece (multi-line FASTA is joined, qualities are gone)
    if fmt != "plain":
        whole = b"".join(got)
        assert whole.count(b">" if fmt == "fasta" else b"@") == len(reads)


def test_irregular_fastq_and_gzip_are_refused(tmp_path):
    p = tmp_path / "ml.fastq"
    p.write_bytes(b"@r0\nACGT\nACGT\n+\nIIIIIIII\n@r1\nAC\n+\nII\n")
    with pytest.raises(ValueError):
        seqstream.to_stream(seqstream.load_bytes(str(p), 0, p.stat().st_size, "cpu"), "fastq")
    g = tmp_path / "x.fa.gz"
    g.write_bytes(b"\x1f\x8b\x08\x00")
    with pytest.raises(ValueError):
        seqstream.detect_format(str(g))


def test_long_record_is_an_error(tmp_path):
    p = tmp_path / "chr.fa"
    p.write_bytes(b">chr\n" + b"ACGT" * 100000 + b"\n>b\nACGT\n")
    with pytest.raises(ValueError):
        seqstream.plan_steps(str(p), "fasta", 0, 1, 4096)
