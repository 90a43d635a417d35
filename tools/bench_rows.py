#!/usr/bin/env python3
"""Throughput of the rows next to the build path on one MI355X:
record bulk load (host buffer -> table, PCIe included), device sort of .ctx records, table scans.
Prints one JSON object; the roofline for the scans is HBM (table bytes / time)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
K, SLOTS = 31, 1 << 28
genome = bench.make_genome(100_000_000, dev, 7)
batches = [bench.make_batch(genome, 5_000_000, 50 + i, dev) for i in range(2)]
del genome
g = mcx.Graph(K, 1, SLOTS)
for b in batches:
    g.add_stream_dev(0, b, b.numel())
g.sync()
n = g.nkmers
out = {"kmer_size": K, "table_slots": SLOTS, "records": n, "record_bytes": 13}

t0 = time.perf_counter(); body = g.export(True); dt = time.perf_counter() - t0
out["export_sorted"] = {"seconds": dt, "records_per_s": n / dt, "GB_per_s": len(body) / dt / 1e9}

for name, fn in (("kmer_covg", lambda: g.kmer_covg()), ("covg_histogram_4096", lambda: g.covg_histogram(4096))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    dt = (time.perf_counter() - t0) / 5
    tb = SLOTS * 16
    out[name] = {"seconds": dt, "table_GB_per_s": tb / dt / 1e9, "hbm_frac": tb / dt / 1e9 / 8000.0}

rec = np.frombuffer(body, np.uint8).reshape(-1, 13)
rng = np.random.default_rng(1)
shuf = rec[rng.permutation(len(rec))].tobytes()
t0 = time.perf_counter(); srt = mcx.sort_records(shuf, K, 1); dt = time.perf_counter() - t0
assert srt == body
out["sort_records"] = {"seconds": dt, "records_per_s": n / dt, "GB_per_s": len(body) / dt / 1e9, "note": "host buffer in and out (PCIe both ways)"}

g2 = mcx.Graph(K, 1, SLOTS)
t0 = time.perf_counter(); st = g2.add_records(shuf, 1, [(0, 0)]); dt = time.perf_counter() - t0
assert st.nkmers_loaded == n and g2.nkmers == n
out["add_records"] = {"seconds": dt, "records_per_s": n / dt, "GB_per_s": len(body) / dt / 1e9, "note": "pageable host buffer -> pinned staging -> device"}
t0 = time.perf_counter(); st = g2.add_records(shuf, 1, [(0, 0)]); dt = time.perf_counter() - t0
out["add_records_existing"] = {"seconds": dt, "records_per_s": n / dt}
assert g2.export(True) != body  # coverage doubled
print(json.dumps(out))
