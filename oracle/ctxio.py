"""oracle/ctxio.py -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Python restatement of the reference's `.ctx` header reader / writer, colour filters and GraphInfo
arithmetic, used to check `build --graph` and the header the host program writes:

  GraphInfo            src/basic/graph_info.c:60-175 (init, update_contigs, merge, cleaning merge)
  parse_filter         src/basic/file_filter.c:7-143 + src/basic/range.c
  read_header/records  src/graph/graph_file_reader.c:78-260,347-420
  header_bytes         src/graph/graph_writer.c:11-110
  load_into            src/graph/graphs_load.c:48-214 (ginfo merge + per-record load)

seq_err is a C `long double`: numpy.longdouble is the same 80-bit x87 type on x86-64.
Parity: unpinned against a reference binary (the reference cannot be built here, oracle/README.md);
pinned against the C oracle's own header writer for graphs built from reads (tests/test_ctxio.py).
"""
import struct

import numpy as np

LD = np.longdouble
assert np.dtype(LD).itemsize == 16, "x87 long double layout expected"


class Cleaning:
    def __init__(self):
        self.cleaned_tips = self.cleaned_unitigs = self.cleaned_kmers = 0
        self.clean_unitigs_thresh = self.clean_kmers_thresh = 0
        self.is_graph_intersection = 0
        self.intersection_name = "undefined"

    def merge(self, src):  # error_cleaning_merge
        self.cleaned_tips |= src.cleaned_tips
        self.cleaned_unitigs |= src.cleaned_unitigs
        self.cleaned_kmers |= src.cleaned_kmers
        if src.clean_unitigs_thresh > 0 and (self.clean_unitigs_thresh == 0 or src.clean_unitigs_thresh < self.clean_unitigs_thresh):
            self.clean_unitigs_thresh = src.clean_unitigs_thresh
        if src.clean_kmers_thresh > 0 and (self.clean_kmers_thresh == 0 or src.clean_kmers_thresh < self.clean_kmers_thresh):
            self.clean_kmers_thresh = src.clean_kmers_thresh
        if src.is_graph_intersection:  # graph_info_append_intersect
            if not self.is_graph_intersection:
                self.intersection_name = src.intersection_name
            else:
                self.intersection_name += "," + src.intersection_name
            self.is_graph_intersection = 1
        self.is_graph_intersection |= src.is_graph_intersection


class GraphInfo:
    def __init__(self):  # graph_info_init
        self.sample_name = "undefined"
        self.total_sequence = 0
        self.mean_read_length = 0
        self.seq_err = LD(0.01)  # the double constant 0.01 widened
        self.cleaning = Cleaning()

    def update_contigs(self, added_seq, num_contigs):  # graph_info_update_contigs
        if not added_seq and not num_contigs:
            return
        have = 0
        if self.total_sequence and self.mean_read_length:
            have = int(float(self.total_sequence) / self.mean_read_length + 0.5)
        if have + num_contigs > 0:
            self.mean_read_length = int(float(self.total_sequence + added_seq) / (have + num_contigs)) & 0xFFFFFFFF
        self.total_sequence += added_seq

    def merge(self, src):  # graph_info_merge
        if src.sample_name != "undefined":
            if self.sample_name == "undefined":
                self.sample_name = src.sample_name
            else:
                self.sample_name += "," + src.sample_name
        total = self.total_sequence + src.total_sequence
        if total > 0:
            self.seq_err = (self.seq_err * LD(self.total_sequence) + src.seq_err * LD(src.total_sequence)) / LD(total)
            src_contigs = 0
            if src.total_sequence and src.mean_read_length:
                src_contigs = int(float(src.total_sequence) / src.mean_read_length + 0.5)
            self.update_contigs(src.total_sequence, src_contigs)
        self.cleaning.merge(src.cleaning)
        self.total_sequence = total


def words_for_k(k):
    return (2 * k + 63) // 64


def header_bytes(k, ginfos):
    """graph_writer_mkhdr + graph_write_header: every colour merged into a fresh GraphInfo"""
    hs = []
    for g in ginfos:
        h = GraphInfo()
        h.merge(g)
        hs.append(h)
    out = b"CORTEX" + struct.pack("<IIII", 6, k, words_for_k(k), len(hs))
    out += b"".join(struct.pack("<I", h.mean_read_length) for h in hs)
    out += b"".join(struct.pack("<Q", h.total_sequence) for h in hs)
    for h in hs:
        n = h.sample_name.encode()
        out += struct.pack("<I", len(n)) + n
    for h in hs:
        out += np.array([h.seq_err], dtype=LD).tobytes()[:10] + b"\0" * 6
    for h in hs:
        c = h.cleaning
        out += bytes([c.cleaned_tips, c.cleaned_unitigs, c.cleaned_kmers, c.is_graph_intersection])
        out += struct.pack("<II", c.clean_unitigs_thresh if c.cleaned_unitigs else 0, c.clean_kmers_thresh if c.cleaned_kmers else 0)
        n = c.intersection_name.encode()
        out += struct.pack("<I", len(n)) + n
    return out + b"CORTEX"


class CtxError(Exception):
    pass


def read_header(buf):
    """-> (dict(version, kmer_size, num_words, num_cols, ginfo[]), header size)"""
    p = 0

    def take(n, what):
        nonlocal p
        if p + n > len(buf):
            raise CtxError("Unexpected end of file [%s]" % what)
        b = buf[p:p + n]
        p += n
        return b

    if take(6, "Magic word") != b"CORTEX":
        raise CtxError("Magic word doesn't match 'CORTEX' (start)")
    version, k, W, ncols = struct.unpack("<IIII", take(16, "header"))
    if version > 7 or version < 4:
        raise CtxError("Sorry, we only support graph file versions 4, 5, 6 & 7")
    if k % 2 == 0:
        raise CtxError("kmer size is not an odd number")
    if k < 3:
        raise CtxError("kmer size is less than three")
    if W * 32 < k:
        raise CtxError("Not enough bitfields for kmer size")
    if (W - 1) * 32 >= k:
        raise CtxError("using more than the minimum number of bitfields")
    if ncols == 0:
        raise CtxError("number of colours is zero")
    if ncols > 10000:
        raise CtxError("Very high number of colours")
    gi = [GraphInfo() for _ in range(ncols)]
    for g in gi:
        g.mean_read_length, = struct.unpack("<I", take(4, "mean read length"))
    for g in gi:
        g.total_sequence, = struct.unpack("<Q", take(8, "total sequence"))
    if version >= 6:
        def name(what):
            n, = struct.unpack("<I", take(4, what))
            if n > 10000:
                raise CtxError("Very big sample name")
            return take(n, what).split(b"\0")[0].decode()
        for g in gi:
            g.sample_name = name("sample name")
        for g in gi:
            g.seq_err = np.frombuffer(take(16, "seq error rates"), dtype=LD)[0]
        for g in gi:
            c = g.cleaning
            c.cleaned_tips, c.cleaned_unitigs, c.cleaned_kmers, c.is_graph_intersection = take(4, "cleaning flags")
            tu, tk = struct.unpack("<II", take(8, "cleaning thresholds"))
            if version <= 6:
                if not c.cleaned_unitigs and tu == 0xFFFFFFFF:
                    tu = 0
                if not c.cleaned_kmers and tk == 0xFFFFFFFF:
                    tk = 0
            if not c.cleaned_unitigs:
                tu = 0
            if not c.cleaned_kmers:
                tk = 0
            c.clean_unitigs_thresh, c.clean_kmers_thresh = tu, tk
            c.intersection_name = name("cleaned against graph name")
    if take(6, "magic word (end)") != b"CORTEX":
        raise CtxError("Magic word doesn't match 'CORTEX' (end)")
    return dict(version=version, kmer_size=k, num_words=W, num_cols=ncols, ginfo=gi), p


def records(buf, hdr, hdr_size):
    """-> keys [n, W] u64, covgs [n, ncols] u32, edges [n, ncols] u8 (whole records only)"""
    W, nc = hdr["num_words"], hdr["num_cols"]
    rs = 8 * W + 5 * nc
    n = (len(buf) - hdr_size) // rs
    a = np.frombuffer(buf, dtype=np.uint8, count=n * rs, offset=hdr_size).reshape(n, rs)
    keys = a[:, :8 * W].copy().view("<u8").reshape(n, W)
    covgs = a[:, 8 * W:8 * W + 4 * nc].copy().view("<u4").reshape(n, nc)
    edges = a[:, 8 * W + 4 * nc:].copy()
    return keys, covgs, edges


# ---- ranges and filters -------------------------------------------------------------------
def _range_parse(s, range_max):
    if s[:1] == "*":
        return 0, range_max, 1
    i = 0
    while i < len(s) and s[i].isdigit():
        i += 1
    if i == 0:
        return None
    frm = to = int(s[:i])
    if s[i:i + 1] == "-":
        j = i + 1
        while j < len(s) and s[j].isdigit():
            j += 1
        if j == i + 1:
            return None
        to = int(s[i + 1:j])
        i = j
    if frm > range_max or to > range_max:
        return None
    return frm, to, i


def range_array(s, range_max):
    """range_parse_array; None on syntax error.  (A descending range a-b lists a, a-1, .. b.)"""
    out, p = [], 0
    while p < len(s):
        r = _range_parse(s[p:], range_max)
        if r is None:
            return None
        a, b, n = r
        p += n
        if s[p:p + 1] == ",":
            p += 1
        out += list(range(a, b + 1)) if a <= b else list(range(a, b - 1, -1))
    if s.endswith(","):
        return None
    if not out:
        out = list(range(range_max + 1))
    return out


def parse_filter(inp, srcncols, into_offset):
    """file_filter_open + file_filter_set_cols -> (path, [(from, into)] sorted by into)"""
    rc = set("0123456789-,")
    i = 0
    while i < len(inp) and inp[i] in rc:
        i += 1
    start = i + 1 if (i > 0 and inp[i:i + 1] == ":") else 0
    end = len(inp)
    p = len(inp)
    while p > start + 1:
        p -= 1
        if inp[p] == ":":
            end = p
            break
        if inp[p] not in rc:
            break
    path = inp[start:end]
    from_f = inp[end + 1:] if inp[end:end + 1] == ":" else None
    into_f = inp[:start - 1] if start > 0 else None
    if from_f is not None:
        frm = range_array(from_f, srcncols - 1)
        if frm is None:
            raise CtxError("Invalid filter path: %s" % inp)
    else:
        frm = list(range(srcncols))
    n = len(frm)
    if into_f is not None:
        into = range_array(into_f, (1 << 63))
        if into is None or (len(into) != 1 and len(into) != n):
            raise CtxError("Invalid filter path: %s" % inp)
        if len(into) == 1:
            into = into * n
    else:
        into = [into_offset + j for j in range(n)]
    return path, sorted(zip(frm, into), key=lambda t: (t[1], t[0]))


def load_into(ograph, ginfos, buf, filt, must_exist=False):
    """graph_load: merge the file's GraphInfo into ginfos[into] and add every record to the C oracle
    graph `ograph` (oracle.orc.Graph).  Returns (read, loaded)."""
    hdr, hs = read_header(buf)
    for frm, into in filt:
        ginfos[into].merge(hdr["ginfo"][frm])
    keys, covgs, edges = records(buf, hdr, hs)
    ncols = ograph.ncols
    loaded = 0
    for i in range(len(keys)):
        cv = [0] * ncols
        ed = [0] * ncols
        for frm, into in filt:  # graph_file_read: SAFE_ADD_COVG + OR
            cv[into] = min(0xFFFFFFFF, cv[into] + int(covgs[i, frm]))
            ed[into] |= int(edges[i, frm])
        loaded += ograph.add_record(keys[i], cv, ed, must_exist) == 1
    return len(keys), loaded
