"""ctypes mirror of include/mcx_gpu.h (one Python method per C entry point)."""
import ctypes as C
import os

import numpy as np

from . import _build

MCX_OK, MCX_ERR_ARG, MCX_ERR_NODEVICE, MCX_ERR_NOMEM, MCX_ERR_FULL, MCX_ERR_HIP, MCX_ERR_SINK = 0, -1, -2, -3, -4, -5, -6

_LIB = None

SYMBOLS = [
    "mcx_last_error", "mcx_version", "mcx_device_count", "mcx_device_memory", "mcx_graph_create", "mcx_graph_create_multi", "mcx_graph_ndevices",
    "mcx_graph_create_shard", "mcx_graph_shard_layout", "mcx_graph_shard_bins_dev", "mcx_graph_add_segments_dev",
    "mcx_graph_key_owner", "mcx_graph_insert_tuple_segments_dev", "mcx_graph_add_records", "mcx_graph_kmer_covg", "mcx_graph_covg_histogram", "mcx_sort_records",
    "mcx_records_sorted", "mcx_graph_intersect_finish", "mcx_superk_supported", "mcx_superk_record_bytes", "mcx_superk_owner", "mcx_graph_checksum", "mcx_records_checksum",
    "mcx_graph_superk_layout", "mcx_graph_superk_bins_dev", "mcx_graph_add_superk_dev", "mcx_graph_destroy",
    "mcx_graph_reset", "mcx_graph_configure", "mcx_graph_profile", "mcx_graph_capacity", "mcx_graph_add_reads", "mcx_graph_add_reads_pcr", "mcx_graph_pcr_reset", "mcx_graph_add_stream_dev",
    "mcx_graph_partition_stream_dev", "mcx_graph_insert_tuples_dev", "mcx_key_owner", "mcx_graph_sync",
    "mcx_graph_nkmers", "mcx_graph_device_stats", "mcx_graph_stream", "mcx_graph_export",
    "mcx_kmer_from_str", "mcx_kmer_canonical", "mcx_kmer_hash", "mcx_pack_bases", "mcx_pack_reads_host", "mcx_pack_stream_dev", "mcx_graph_add_packed_dev",
    "mcx_ubench_stream", "mcx_ubench_random_rmw", "mcx_graph_insert_stats", "mcx_multi_exchange_bytes", "mcx_graph_hashtest", "mcx_hashtest_func", "mcx_debug_probe",
]


class McxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mcx error %d: %s" % (code, msg))
        self.code = code


class LoadStats(C.Structure):
    """mcx_load_stats (subset of the reference's SeqLoadingStats)."""
    _fields_ = [(n, C.c_uint64) for n in (
        "num_se_reads", "num_good_reads", "num_bad_reads", "total_bases_read",
        "total_bases_loaded", "contigs_parsed", "num_kmers_loaded", "num_kmers_novel",
        "num_pe_reads", "num_dup_se_reads", "num_dup_pe_pairs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RecordStats(C.Structure):
    """mcx_records_stats"""
    _fields_ = [("nkmers_read", C.c_uint64), ("nkmers_loaded", C.c_uint64), ("nkmers_novel", C.c_uint64),
                ("first_oversized", C.c_int64), ("first_zero_covg", C.c_int64), ("first_edges_no_covg", C.c_int64)]

    def __init__(self):
        super().__init__(0, 0, 0, -1, -1, -1)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RECORDS_MUST_EXIST = 1
SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


def lib():
    """Load libmcxgpu.so; raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  Importing torch first makes the loader bind this library to the runtime
    # torch uses, so torch tensors' device pointers are valid in our kernels.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("MCX_LIB") or _build.LIB  # MCX_LIB: tuning variants built by tools/
    if not os.path.exists(path):
        raise RuntimeError("HIP extension missing: %s (run __graft_entry__.build())" % path)
    L = C.CDLL(path)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.mcx_last_error.restype = C.c_char_p
    L.mcx_version.restype = C.c_char_p
    L.mcx_device_count.restype = C.c_int
    L.mcx_graph_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint64, C.c_int]
    L.mcx_graph_create_shard.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int]
    L.mcx_graph_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.c_int]
    L.mcx_graph_ndevices.argtypes = [vp]
    L.mcx_graph_shard_layout.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), u64p, u64p]
    L.mcx_graph_shard_bins_dev.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64]
    L.mcx_graph_add_segments_dev.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, C.c_uint64, C.c_uint64]
    L.mcx_graph_insert_tuple_segments_dev.argtypes = [vp, C.c_int, vp, vp, vp, C.c_uint32, C.c_uint64]
    L.mcx_graph_add_records.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_int, C.c_uint32, C.POINTER(RecordStats)]
    L.mcx_graph_kmer_covg.argtypes = [vp, u64p, u64p]
    L.mcx_graph_insert_stats.argtypes = [vp, vp]
    L.mcx_graph_covg_histogram.argtypes = [vp, u64p, C.c_uint32]
    L.mcx_sort_records.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_int]
    L.mcx_records_sorted.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.mcx_graph_intersect_finish.argtypes = [vp, u64p]
    L.mcx_superk_supported.argtypes = [C.c_int]
    L.mcx_superk_record_bytes.argtypes = [C.c_int]
    L.mcx_superk_owner.restype = C.c_uint32
    L.mcx_superk_owner.argtypes = [u64p, C.c_int, C.c_int]
    L.mcx_graph_superk_layout.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(C.c_uint32), u64p]
    L.mcx_graph_superk_bins_dev.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_uint64]
    L.mcx_graph_add_superk_dev.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, C.c_uint64, C.c_uint64]
    L.mcx_graph_checksum.argtypes = [vp, u64p, u64p]
    L.mcx_records_checksum.restype = C.c_uint64
    L.mcx_records_checksum.argtypes = [vp, C.c_uint64, C.c_int, C.c_int]
    L.mcx_ubench_stream.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mcx_ubench_random_rmw.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mcx_graph_key_owner.restype = C.c_uint32
    L.mcx_graph_key_owner.argtypes = [vp, u64p]
    L.mcx_graph_destroy.argtypes = [vp]
    L.mcx_graph_destroy.restype = None
    L.mcx_graph_reset.argtypes = [vp]
    L.mcx_graph_capacity.argtypes = [vp, u64p, u64p]
    L.mcx_graph_configure.argtypes = [vp, C.c_char_p, C.c_uint64]
    L.mcx_graph_profile.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.mcx_graph_add_reads.argtypes = [vp, C.c_int, vp, vp, vp, C.c_uint64, C.c_uint8, C.c_uint8,
                                      C.POINTER(LoadStats)]
    L.mcx_graph_add_reads_pcr.argtypes = [vp, C.c_int, vp, vp, vp, C.c_uint64, C.c_uint8, C.c_uint8, C.c_uint8,
                                          C.c_int, C.c_int, C.POINTER(LoadStats)]
    L.mcx_graph_pcr_reset.argtypes = [vp]
    L.mcx_graph_add_stream_dev.argtypes = [vp, C.c_int, vp, C.c_uint64]
    L.mcx_pack_stream_dev.argtypes = [vp, C.c_uint64, vp, vp, vp]
    L.mcx_graph_add_packed_dev.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.mcx_graph_partition_stream_dev.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_uint64, vp, vp, vp]
    L.mcx_graph_insert_tuples_dev.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.mcx_graph_hashtest.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.mcx_hashtest_func.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    L.mcx_key_owner.restype = C.c_uint32
    L.mcx_key_owner.argtypes = [u64p, C.c_int, C.c_int]
    L.mcx_graph_sync.argtypes = [vp]
    L.mcx_graph_nkmers.argtypes = [vp, u64p]
    L.mcx_graph_device_stats.argtypes = [vp, C.POINTER(LoadStats)]
    L.mcx_graph_stream.restype = vp
    L.mcx_graph_stream.argtypes = [vp]
    L.mcx_graph_export.argtypes = [vp, C.c_int, SINK_FN, vp]
    L.mcx_kmer_from_str.restype = None
    L.mcx_kmer_from_str.argtypes = [C.c_char_p, C.c_int, u64p]
    L.mcx_kmer_canonical.restype = None
    L.mcx_kmer_canonical.argtypes = [u64p, C.c_int, u64p, C.POINTER(C.c_int)]
    L.mcx_kmer_hash.restype = C.c_uint32
    L.mcx_kmer_hash.argtypes = [u64p, C.c_int, C.c_uint32]
    _LIB = L
    return L


def _check(rc):
    if rc != MCX_OK:
        raise McxError(rc, lib().mcx_last_error().decode())


def device_count():
    return lib().mcx_device_count()


def _words(k):
    return (2 * k + 63) // 64


def kmer_from_str(s, k):
    out = (C.c_uint64 * 4)()
    lib().mcx_kmer_from_str(s.encode() if isinstance(s, str) else s, k, out)
    return [int(out[i]) for i in range(_words(k))]


def kmer_canonical(words, k):
    a = (C.c_uint64 * 4)(*words)
    out = (C.c_uint64 * 4)()
    o = C.c_int()
    lib().mcx_kmer_canonical(a, k, out, C.byref(o))
    return [int(out[i]) for i in range(_words(k))], int(o.value)


def kmer_hash(words, k, initval=0):
    a = (C.c_uint64 * 4)(*words)
    return int(lib().mcx_kmer_hash(a, k, initval))


def key_owner(words, k, nparts):
    a = (C.c_uint64 * 2)(*words)
    return int(lib().mcx_key_owner(a, k, nparts))


def stream_from_reads(reads):
    """Host helper: reads -> the separator-delimited byte stream the device entry takes."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    return np.frombuffer(b"".join(b + b"\n" for b in bs), dtype=np.uint8).copy()


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if hasattr(x, "data_ptr"):  # torch tensor (device or host)
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Graph:
    """One coloured de Bruijn graph resident in the HBM of one GPU."""

    def __init__(self, kmer_size, ncols=1, capacity=1 << 20, device=0, nparts=1, part=0, devices=None):
        """devices: list of device ordinals -> one table split over them (mcx_graph_create_multi);
        the same ordinal may repeat (several shards on one GPU: how the path is tested on one)."""
        self.L = lib()
        self.k, self.ncols, self.W = kmer_size, ncols, _words(kmer_size)
        self.nparts, self.part = nparts, part
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _check(self.L.mcx_graph_create_multi(C.byref(h), kmer_size, ncols, capacity, arr, len(devices)))
        else:
            _check(self.L.mcx_graph_create_shard(C.byref(h), kmer_size, ncols, capacity, device, nparts, part))
        self.h = h

    @property
    def ndevices(self):
        return int(self.L.mcx_graph_ndevices(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.mcx_graph_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        _check(self.L.mcx_graph_reset(self.h))

    def configure(self, key, value):
        _check(self.L.mcx_graph_configure(self.h, key.encode(), int(value)))

    def profile(self):
        """{kernel: (calls, total_ms)} for launches recorded since configure('profile', 1)."""
        buf = C.create_string_buffer(16384)   # (a multi-GPU handle reports every shard's kernels: name@shard)
        _check(self.L.mcx_graph_profile(self.h, buf, 16384))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = (int(calls), float(ms))
        return out

    def capacity(self):
        s, b = C.c_uint64(), C.c_uint64()
        _check(self.L.mcx_graph_capacity(self.h, C.byref(s), C.byref(b)))
        return int(s.value), int(b.value)

    def add_reads(self, colour, bases, offsets, quals=None, fq_cutoff=0, hp_cutoff=0, stats=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        st = stats if stats is not None else LoadStats()
        _check(self.L.mcx_graph_add_reads(self.h, colour, _ptr(bases), _ptr(quals), _ptr(offsets),
                                          len(offsets) - 1, fq_cutoff, hp_cutoff, C.byref(st)))
        return st

    MATEDIR = {"FF": 0, "FR": 1, "RF": 2, "RR": 3}

    def add_reads_pcr(self, colour, bases, offsets, quals=None, fq_cutoff=0, fq_cutoff2=None, hp_cutoff=0,
                      paired=False, matedir="FR", stats=None):
        """`build --remove-pcr` over one batch (reads 2i, 2i + 1 are mates when `paired`)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        st = stats if stats is not None else LoadStats()
        _check(self.L.mcx_graph_add_reads_pcr(self.h, colour, _ptr(bases), _ptr(quals), _ptr(offsets), len(offsets) - 1,
                                              fq_cutoff, fq_cutoff if fq_cutoff2 is None else fq_cutoff2, hp_cutoff,
                                              1 if paired else 0, self.MATEDIR.get(matedir, matedir), C.byref(st)))
        return st

    def pcr_reset(self):
        _check(self.L.mcx_graph_pcr_reset(self.h))

    def superk_layout(self, nparts, positions_per_call):
        segs, cap = C.c_uint32(), C.c_uint64()
        _check(self.L.mcx_graph_superk_layout(self.h, nparts, int(positions_per_call), C.byref(segs), C.byref(cap)))
        return int(segs.value), int(cap.value)

    def superk_bins_dev(self, d_stream, nbytes, nparts, d_recs, d_counts, seg_cap):
        _check(self.L.mcx_graph_superk_bins_dev(self.h, _ptr(d_stream), nbytes, nparts, _ptr(d_recs), _ptr(d_counts), seg_cap))

    def add_superk_dev(self, colour, d_recs, d_counts, nseg, seg_cap, kmers_upper_bound):
        _check(self.L.mcx_graph_add_superk_dev(self.h, colour, _ptr(d_recs), _ptr(d_counts), nseg, seg_cap, int(kmers_upper_bound)))

    def checksum(self):
        """(order-independent checksum of the exported records, number of k-mers)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self.L.mcx_graph_checksum(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def intersect_finish(self):
        n = C.c_uint64(0)
        _check(self.L.mcx_graph_intersect_finish(self.h, C.byref(n)))
        return int(n.value)

    def add_records(self, recs, file_ncols, colour_filter, must_exist=False, stats=None, mask_edges=False):
        """recs: .ctx body bytes (W x u64 key, file_ncols x u32 coverage, file_ncols x u8 edges per
        record); colour_filter: [(from file colour, into graph colour)]"""
        recs = np.frombuffer(recs, dtype=np.uint8) if isinstance(recs, (bytes, bytearray, memoryview)) else np.ascontiguousarray(recs, dtype=np.uint8)
        rs = 8 * self.W + 5 * file_ncols
        assert recs.size % rs == 0
        frm = np.array([f for f, _ in colour_filter], dtype=np.int32)
        into = np.array([t for _, t in colour_filter], dtype=np.int32)
        st = stats if stats is not None else RecordStats()
        _check(self.L.mcx_graph_add_records(self.h, _ptr(recs), recs.size // rs, file_ncols, _ptr(frm), _ptr(into),
                                            len(frm), (RECORDS_MUST_EXIST if must_exist else 0) | (2 if mask_edges else 0), C.byref(st)))
        return st

    def kmer_covg(self):
        """per colour: (k-mers with coverage, summed coverage) -- db_graph_get_kmer_covg"""
        nk = np.zeros(self.ncols, dtype=np.uint64)
        sc = np.zeros(self.ncols, dtype=np.uint64)
        _check(self.L.mcx_graph_kmer_covg(self.h, nk.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_uint64))))
        return nk, sc

    def covg_histogram(self, nbins):
        h = np.zeros(nbins, dtype=np.uint64)
        _check(self.L.mcx_graph_covg_histogram(self.h, h.ctypes.data_as(C.POINTER(C.c_uint64)), nbins))
        return h

    def add_packed_dev(self, colour, d_code, d_inv, npos):
        """a device-resident stream in packed form (pack_stream_dev): 3 bits per position"""
        _check(self.L.mcx_graph_add_packed_dev(self.h, colour, _ptr(d_code), _ptr(d_inv), int(npos)))

    def add_stream_dev(self, colour, d_stream, nbytes):
        _check(self.L.mcx_graph_add_stream_dev(self.h, colour, _ptr(d_stream), nbytes))

    def partition_stream_dev(self, d_stream, nbytes, nparts, bin_capacity, d_keys, d_edges, d_counts):
        _check(self.L.mcx_graph_partition_stream_dev(self.h, _ptr(d_stream), nbytes, nparts, bin_capacity,
                                                     _ptr(d_keys), _ptr(d_edges), _ptr(d_counts)))

    def hashtest(self, first, n):
        """The reference's `hashtest`: find-or-insert of the keys whose top word is first .. first + n - 1."""
        _check(self.L.mcx_graph_hashtest(self.h, first, n))

    def insert_tuples_dev(self, colour, d_keys, d_edges, n):
        _check(self.L.mcx_graph_insert_tuples_dev(self.h, colour, _ptr(d_keys), _ptr(d_edges), n))

    def shard_layout(self, tuples_per_call):
        """(segments per owner, tuples per segment, overflow capacity per owner)"""
        segs, cap, ov = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(self.L.mcx_graph_shard_layout(self.h, int(tuples_per_call), C.byref(segs), C.byref(cap), C.byref(ov)))
        return int(segs.value), int(cap.value), int(ov.value)

    def shard_bins_dev(self, d_stream, nbytes, d_keys, d_counts, seg_cap, d_ov_keys, d_ov_edges, d_ov_counts, ov_cap):
        _check(self.L.mcx_graph_shard_bins_dev(self.h, _ptr(d_stream), nbytes, _ptr(d_keys), _ptr(d_counts), seg_cap,
                                               _ptr(d_ov_keys), _ptr(d_ov_edges), _ptr(d_ov_counts), ov_cap))

    def add_segments_dev(self, colour, d_keys, d_counts, nseg, seg_cap, ntuples):
        _check(self.L.mcx_graph_add_segments_dev(self.h, colour, _ptr(d_keys), _ptr(d_counts), nseg, seg_cap, int(ntuples)))

    def insert_tuple_segments_dev(self, colour, d_keys, d_edges, d_counts, nseg, seg_cap):
        _check(self.L.mcx_graph_insert_tuple_segments_dev(self.h, colour, _ptr(d_keys), _ptr(d_edges), _ptr(d_counts),
                                                          nseg, seg_cap))

    def key_owner(self, words):
        a = (C.c_uint64 * 2)(*words)
        return int(self.L.mcx_graph_key_owner(self.h, a))

    def sync(self):
        _check(self.L.mcx_graph_sync(self.h))

    @property
    def nkmers(self):
        n = C.c_uint64()
        _check(self.L.mcx_graph_nkmers(self.h, C.byref(n)))
        return int(n.value)

    def device_stats(self):
        st = LoadStats()
        _check(self.L.mcx_graph_device_stats(self.h, C.byref(st)))
        return st

    def insert_stats(self):
        """mcx_insert_stats as a dict: fallback_inserts, foreign_inserts, spilled, flushes (implies a sync)"""
        st = (C.c_uint64 * 4)()
        _check(self.L.mcx_graph_insert_stats(self.h, st))
        return dict(zip(("fallback_inserts", "foreign_inserts", "spilled", "flushes"), (int(x) for x in st)))

    @property
    def stream(self):
        return self.L.mcx_graph_stream(self.h)

    def export(self, sorted_=True):
        """Records in .ctx body layout as one bytes object."""
        parts = []

        def sink(_ctx, ptr, n):
            parts.append(C.string_at(ptr, n))
            return 0

        cb = SINK_FN(sink)
        _check(self.L.mcx_graph_export(self.h, 1 if sorted_ else 0, cb, None))
        return b"".join(parts)

    def records(self, sorted_=True):
        """(keys[n,W] u64, covg[n,ncols] u32, edges[n,ncols] u8)."""
        body = self.export(sorted_)
        rs = 8 * self.W + 5 * self.ncols
        rec = np.frombuffer(body, dtype=np.uint8).reshape(-1, rs)
        keys = rec[:, :8 * self.W].copy().view(np.uint64).reshape(-1, self.W)
        cov = rec[:, 8 * self.W:8 * self.W + 4 * self.ncols].copy().view(np.uint32).reshape(-1, self.ncols)
        edg = rec[:, 8 * self.W + 4 * self.ncols:].copy()
        return keys, cov, edg


def sort_records(recs, kmer_size, ncols, device=0):
    """sort .ctx body bytes by k-mer on the device; returns the sorted bytes"""
    a = np.frombuffer(bytes(recs), dtype=np.uint8).copy()
    rs = 8 * _words(kmer_size) + 5 * ncols
    assert a.size % rs == 0
    _check(lib().mcx_sort_records(_ptr(a), a.size // rs, kmer_size, ncols, device))
    return a.tobytes()


def records_sorted(recs, kmer_size, ncols, device=0):
    """index of the first record that is not greater than its predecessor, or -1"""
    a = np.frombuffer(bytes(recs), dtype=np.uint8)
    rs = 8 * _words(kmer_size) + 5 * ncols
    bad = C.c_int64(-1)
    _check(lib().mcx_records_sorted(_ptr(a), a.size // rs, kmer_size, ncols, device, C.byref(bad)))
    return int(bad.value)


def superk_supported(kmer_size):
    return bool(lib().mcx_superk_supported(kmer_size))


def multi_exchange_bytes(kmer_size, ndevices, capacity_kmers):
    """HBM per device that the exchange buffers of an N-device table take (mcx_multi_exchange_bytes; no GPU needed)"""
    out = C.c_uint64(0)
    L = lib()
    L.mcx_multi_exchange_bytes.argtypes = [C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]
    rc = L.mcx_multi_exchange_bytes(kmer_size, ndevices, capacity_kmers, C.byref(out))
    if rc != 0:
        raise McxError(rc, L.mcx_last_error().decode())
    return int(out.value)


def superk_record_words(kmer_size):
    """64-bit words per super-k-mer record (2 for one-word keys, 4 for two-word keys)"""
    return int(lib().mcx_superk_record_bytes(kmer_size)) // 8


def superk_owner(words, kmer_size, nparts):
    a = (C.c_uint64 * 2)(*(list(words) + [0])[:2])
    return int(lib().mcx_superk_owner(a, kmer_size, nparts))


def pack_stream_dev(d_stream, nbytes, d_code, d_inv, hip_stream=None):
    """ASCII stream in HBM -> packed form (d_code: int32 [(nbytes + 15) // 16], d_inv: int16 likewise)"""
    _check(lib().mcx_pack_stream_dev(_ptr(d_stream), int(nbytes), _ptr(d_code), _ptr(d_inv), C.c_void_p(hip_stream) if hip_stream else None))


def ubench_stream(nbytes=8 << 30, device=0):
    """measured streaming ceilings of the device in GB/s: {"copy", "read", "write"} (mcx_ubench_stream)"""
    c, r, w = C.c_double(0), C.c_double(0), C.c_double(0)
    _check(lib().mcx_ubench_stream(device, C.c_uint64(nbytes), C.byref(c), C.byref(r), C.byref(w)))
    return {"copy": c.value, "read": r.value, "write": w.value}


def ubench_random_rmw(table_bytes=16 << 30, nupdates=1 << 29, device=0):
    """measured random 64-byte-sector rates per second over a working set of `table_bytes`:
    {"rmw" (agent-scope atomic), "load16", "load_rmw"} (mcx_ubench_random_rmw)"""
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    _check(lib().mcx_ubench_random_rmw(device, C.c_uint64(table_bytes), C.c_uint64(nupdates), C.byref(a), C.byref(b), C.byref(c)))
    return {"rmw": a.value, "load16": b.value, "load_rmw": c.value}


def records_checksum(recs, kmer_size, ncols):
    if isinstance(recs, np.ndarray) and recs.dtype == np.uint8 and recs.flags.c_contiguous:
        a = recs  # (no copy: bodies of hundreds of millions of records)
    else:
        a = np.frombuffer(bytes(recs), dtype=np.uint8)
    rs = 8 * _words(kmer_size) + 5 * ncols
    assert a.size % rs == 0
    return int(lib().mcx_records_checksum(_ptr(a), a.size // rs, kmer_size, ncols))
