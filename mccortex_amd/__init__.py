"""mccortex_amd -- MI355X (gfx950) backend for the McCortex `build` hot path.

The product is the C-ABI library `libmcxgpu.so` (include/mcx_gpu.h) and the C
host program `mccortex<K> build`; this package is the thin ctypes mirror of
that ABI used by the tests and by bench.py.  There is no CPU path: without the
HIP extension or without a GPU every graph operation raises.
"""
from .graph import (Graph, LoadStats, McxError, MCX_ERR_FULL, device_count, lib, kmer_from_str,
                    kmer_canonical, kmer_hash, key_owner, stream_from_reads, RecordStats, sort_records,
                    records_sorted, superk_supported, superk_record_words, superk_owner, records_checksum, pack_stream_dev,
                    ubench_stream, ubench_random_rmw, multi_exchange_bytes)
from .ctx import CtxHeader, ctx_header_bytes, graph_info_update, write_ctx

__all__ = ["Graph", "LoadStats", "McxError", "MCX_ERR_FULL", "device_count", "lib", "kmer_from_str",
           "kmer_canonical", "kmer_hash", "key_owner", "stream_from_reads", "RecordStats", "sort_records",
           "records_sorted", "superk_supported", "superk_record_words", "superk_owner", "records_checksum", "pack_stream_dev", "ubench_stream", "ubench_random_rmw", "multi_exchange_bytes", "CtxHeader",
           "ctx_header_bytes", "graph_info_update", "write_ctx"]
