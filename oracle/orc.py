"""ctypes binding for the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (mccortex_amd/) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

MAX_W = 4


class BKmer(C.Structure):
    _fields_ = [("b", C.c_uint64 * MAX_W)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "num_se_reads", "num_good_reads", "num_bad_reads", "total_bases_read",
        "total_bases_loaded", "contigs_parsed", "num_kmers_loaded", "num_kmers_novel")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile liborc.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "mcx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    u64p = C.POINTER(C.c_uint64)
    L.orc_words_for_k.restype = C.c_int
    L.orc_kmer_from_str.restype = BKmer
    L.orc_kmer_from_str.argtypes = [C.c_char_p, C.c_int]
    L.orc_kmer_shift_add.restype = BKmer
    L.orc_kmer_shift_add.argtypes = [BKmer, C.c_int, C.c_int]
    L.orc_kmer_revcomp.restype = BKmer
    L.orc_kmer_revcomp.argtypes = [BKmer, C.c_int]
    L.orc_kmer_get_key.restype = BKmer
    L.orc_kmer_get_key.argtypes = [BKmer, C.c_int]
    L.orc_kmer_hash.restype = C.c_uint32
    L.orc_kmer_hash.argtypes = [BKmer, C.c_int, C.c_uint32]
    L.orc_hashtest_func.restype = C.c_uint64
    L.orc_hashtest_func.argtypes = [C.c_int, C.c_uint64, C.c_uint32]
    L.orc_kmer_to_str.argtypes = [BKmer, C.c_int, C.c_char_p]
    L.orc_hash_table_cap.restype = C.c_uint64
    L.orc_hash_table_cap.argtypes = [C.c_uint64, u64p, C.POINTER(C.c_uint8)]
    L.orc_contig_start.restype = C.c_size_t
    L.orc_contig_start.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t,
                                   C.c_size_t, C.c_uint8, C.c_uint8]
    L.orc_contig_end.restype = C.c_size_t
    L.orc_contig_end.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t,
                                 C.c_size_t, C.c_uint8, C.c_uint8, C.POINTER(C.c_size_t)]
    L.orc_graph_new.restype = C.c_void_p
    L.orc_graph_new.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32]
    L.orc_graph_free.argtypes = [C.c_void_p]
    L.orc_graph_set_sample.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    L.orc_graph_force_generic.argtypes = [C.c_void_p, C.c_int]
    L.orc_graph_nkmers.restype = C.c_uint64
    L.orc_graph_nkmers.argtypes = [C.c_void_p]
    L.orc_graph_capacity.restype = C.c_uint64
    L.orc_graph_capacity.argtypes = [C.c_void_p]
    L.orc_graph_add_reads.restype = C.c_int
    L.orc_graph_add_reads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_uint8, C.c_uint8, C.c_int, C.POINTER(Stats)]
    L.orc_graph_update_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.orc_graph_tune.argtypes = [C.c_void_p, C.c_int]
    L.orc_graph_tune.restype = None
    L.orc_build_file.restype = C.c_double
    L.orc_build_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(Stats)]
    L.orc_graph_ctx_size.restype = C.c_size_t
    L.orc_graph_ctx_size.argtypes = [C.c_void_p]
    L.orc_graph_header_size.restype = C.c_size_t
    L.orc_graph_header_size.argtypes = [C.c_void_p]
    L.orc_graph_write_ctx.restype = C.c_size_t
    L.orc_graph_write_ctx.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_graph_lookup.restype = C.c_int
    L.orc_graph_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]
    L.orc_graph_add_record.restype = C.c_int
    L.orc_graph_add_record.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_graph_set_must_exist.argtypes = [C.c_void_p, C.c_int]
    L.orc_graph_add_isec_record.restype = C.c_int
    L.orc_graph_add_isec_record.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint8]
    L.orc_graph_isec_finish.argtypes = [C.c_void_p]
    L.orc_graph_add_reads_pcr.restype = C.c_int
    L.orc_graph_add_reads_pcr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_int, C.POINTER(Stats), C.c_void_p]
    L.orc_graph_pcr_reset.argtypes = [C.c_void_p]
    L.orc_tuples.restype = C.c_uint64
    L.orc_tuples.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                             C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p]
    _LIB = L
    return L


def ref_lookup3():
    """The reference's own libs/misc/lookup3.h, compiled into oracle/_ref (or None)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "liblk3ref.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
        _REF.ref_lk3_hashlittle.restype = C.c_uint32
        _REF.ref_lk3_hashlittle.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    return _REF


_HASHMEM = None


def ref_hashmem():
    """The reference's self-contained src/basic/hash_mem.h (constants + ht_mem) compiled into oracle/_ref (or None)."""
    global _HASHMEM
    if _HASHMEM is None:
        so = os.path.join(_HERE, "_ref", "libhashmemref.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_ht_mem.restype = C.c_size_t
        R.ref_ht_mem.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
        R.ref_ideal_occupancy.restype = C.c_float
        R.ref_warn_occupancy.restype = C.c_float
        _HASHMEM = R
    return _HASHMEM


_REVCMP = {}


def ref_revcmp(W):
    """The reference's standalone dev/bkmer_revcmp/revcmp.c compiled with NUM_BKMER_WORDS = W (1 or 2)
    into oracle/_ref (or None): ref_revcmp(method 1..4, in words, k, out words)."""
    if W not in _REVCMP:
        so = os.path.join(_HERE, "_ref", "librevcmp%d.so" % W)
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_revcmp_words.restype = C.c_int
        R.ref_revcmp.restype = None
        R.ref_revcmp.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        assert R.ref_revcmp_words() == W
        if hasattr(R, "ref_bklk3_hashlittle"):   # src/kmer/kmer_hash.h compiled in as well (round 5)
            R.ref_bklk3_hashlittle.restype = C.c_uint32
            R.ref_bklk3_hashlittle.argtypes = [C.c_void_p, C.c_uint32]
        _REVCMP[W] = R
    return _REVCMP[W]


def pack_reads(reads):
    """list of bytes/str -> (bases uint8[total], offsets uint64[n+1]) without separators."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return bases, offs


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Graph:
    """Oracle graph: reference-shaped bucketed table + covg/edge arrays."""

    def __init__(self, k, ncols=1, capacity=1 << 20, seed=12345):
        self.L = lib()
        self.k, self.ncols = k, ncols
        self.W = self.L.orc_words_for_k(k)
        self.h = self.L.orc_graph_new(k, ncols, capacity, seed)
        if not self.h:
            raise ValueError("bad oracle graph parameters")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_graph_free(self.h)
            self.h = None

    def tune(self, nthreads):
        """timed baseline only: prefault the arrays with `nthreads` threads, huge pages, per-worker node tallies"""
        self.L.orc_graph_tune(self.h, nthreads)

    def build_file(self, path, nthreads):
        """reference-shaped build of one FASTQ / FASTA file (1 reader thread + nthreads workers) -> (total s, insert s, stats)"""
        st, ins = Stats(), C.c_double(0)
        tot = self.L.orc_build_file(self.h, path.encode(), nthreads, C.byref(ins), C.byref(st))
        if tot < 0:
            raise RuntimeError("cannot open %s" % path)
        return tot, ins.value, st

    def force_generic(self, on=True):
        self.L.orc_graph_force_generic(self.h, 1 if on else 0)

    def set_sample(self, col, name):
        assert self.L.orc_graph_set_sample(self.h, col, name.encode()) == 0

    def add_reads(self, colour, bases, offsets, quals=None, fq_cutoff=0, hp_cutoff=0, nthreads=1,
                  stats=None):
        st = stats if stats is not None else Stats()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rc = self.L.orc_graph_add_reads(self.h, colour, _ptr(bases), _ptr(quals), _ptr(offsets),
                                        len(offsets) - 1, fq_cutoff, hp_cutoff, nthreads, C.byref(st))
        if rc != 0:
            raise RuntimeError("Hash table is full" if rc == -1 else "oracle error %d" % rc)
        return st

    MATEDIR = {"FF": 0, "FR": 1, "RF": 2, "RR": 3}

    def add_reads_pcr(self, colour, bases, offsets, quals=None, fq_cutoff=0, fq_cutoff2=None, hp_cutoff=0,
                      paired=False, matedir="FR", stats=None):
        """build --remove-pcr over a batch, in read order; returns (stats, [dup SE reads, dup pairs, PE reads])"""
        st = stats if stats is not None else Stats()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        counts = np.zeros(3, dtype=np.uint64)
        rc = self.L.orc_graph_add_reads_pcr(self.h, colour, _ptr(bases), _ptr(quals), _ptr(offsets), len(offsets) - 1,
                                            fq_cutoff, fq_cutoff if fq_cutoff2 is None else fq_cutoff2, hp_cutoff,
                                            1 if paired else 0, self.MATEDIR.get(matedir, matedir), C.byref(st), _ptr(counts))
        if rc != 0:
            raise RuntimeError("Hash table is full" if rc == -1 else "oracle error %d" % rc)
        return st, [int(x) for x in counts]

    def pcr_reset(self):
        self.L.orc_graph_pcr_reset(self.h)

    def add_record(self, key_words, covgs, edges, must_exist=False):
        """graph_load's per-record body: colours already mapped onto this graph's (ncols entries)"""
        kw = np.ascontiguousarray(key_words, dtype=np.uint64)
        cv = np.ascontiguousarray(covgs, dtype=np.uint32)
        ed = np.ascontiguousarray(edges, dtype=np.uint8)
        assert len(kw) == self.W and len(cv) == self.ncols and len(ed) == self.ncols
        rc = self.L.orc_graph_add_record(self.h, _ptr(kw), _ptr(cv), _ptr(ed), 1 if must_exist else 0)
        if rc < 0:
            raise RuntimeError("Hash table is full")
        return rc

    def set_must_exist(self, on=True):
        self.L.orc_graph_set_must_exist(self.h, 1 if on else 0)

    def add_isec_record(self, key_words, covg_sum, edges_or):
        kw = np.ascontiguousarray(key_words, dtype=np.uint64)
        rc = self.L.orc_graph_add_isec_record(self.h, _ptr(kw), int(covg_sum), int(edges_or))
        if rc < 0:
            raise RuntimeError("Hash table is full")
        return rc

    def isec_finish(self):
        self.L.orc_graph_isec_finish(self.h)

    def body_bytes(self, sorted_=True):
        return self.ctx_bytes(sorted_)[self.header_size():]

    def update_stats(self, colour, st):
        self.L.orc_graph_update_stats(self.h, colour, C.byref(st))

    @property
    def nkmers(self):
        return int(self.L.orc_graph_nkmers(self.h))

    def ctx_bytes(self, sorted_=True):
        n = self.L.orc_graph_ctx_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        w = self.L.orc_graph_write_ctx(self.h, 1 if sorted_ else 0, _ptr(buf))
        assert w == n, (w, n)
        return buf.tobytes()

    def header_size(self):
        return int(self.L.orc_graph_header_size(self.h))

    def body_array(self, sorted_=False):
        """the records of the .ctx (no header) as a uint8 array, without the extra copies of ctx_bytes():
        for graphs of hundreds of millions of k-mers (full-size parity checks, bench.py's cpu_baseline)"""
        n = self.L.orc_graph_ctx_size(self.h)
        buf = np.empty(n, dtype=np.uint8)
        w = self.L.orc_graph_write_ctx(self.h, 1 if sorted_ else 0, _ptr(buf))
        assert w == n, (w, n)
        return buf[self.header_size():]

    def lookup(self, kmer):
        cov = np.zeros(self.ncols, np.uint32)
        edg = np.zeros(self.ncols, np.uint8)
        ok = self.L.orc_graph_lookup(self.h, kmer.encode(), _ptr(cov), _ptr(edg))
        return (cov, edg) if ok else None


def tuples(k, bases, offsets, quals=None, fq_cutoff=0, hp_cutoff=0):
    """Per-occurrence (key words, edge byte) stream in read order."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nreads = len(offsets) - 1
    n = L.orc_tuples(k, _ptr(bases), _ptr(quals), _ptr(offsets), nreads, fq_cutoff, hp_cutoff, None, None)
    W = L.orc_words_for_k(k)
    keys = np.zeros((n, W), dtype=np.uint64)
    edges = np.zeros(n, dtype=np.uint8)
    L.orc_tuples(k, _ptr(bases), _ptr(quals), _ptr(offsets), nreads, fq_cutoff, hp_cutoff, _ptr(keys), _ptr(edges))
    return keys, edges


def graph_from_tuples(keys, edges, colours=None, ncols=1):
    """Reduce a tuple stream to the sorted record set {key: (covg[], edges[])} with numpy --
    the order-independent definition of the graph."""
    W = keys.shape[1]
    if colours is None:
        colours = np.zeros(len(keys), dtype=np.int64)
    order = np.lexsort([keys[:, w] for w in range(W - 1, -1, -1)])
    ks = keys[order]
    es = edges[order]
    cs = np.asarray(colours)[order]
    if len(ks) == 0:
        return ks, np.zeros((0, ncols), np.uint32), np.zeros((0, ncols), np.uint8)
    new = np.ones(len(ks), dtype=bool)
    new[1:] = np.any(ks[1:] != ks[:-1], axis=1)
    gid = np.cumsum(new) - 1
    ng = int(gid[-1]) + 1
    cov = np.zeros((ng, ncols), dtype=np.uint64)
    np.add.at(cov, (gid, cs), 1)
    edg = np.zeros((ng, ncols), dtype=np.uint8)
    np.bitwise_or.at(edg, (gid, cs), es)
    return ks[new], np.minimum(cov, 0xFFFFFFFF).astype(np.uint32), edg
