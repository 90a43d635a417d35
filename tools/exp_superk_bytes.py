#!/usr/bin/env python3
"""Exchange format v3: bytes per occurrence and per-owner balance of the super-k-mer records for
N = 2, 4, 8, 16 shards on one bench batch (5 M x 150 bp), and the sender kernel's time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
batch = bench.make_batch(genome, 5_000_000, 1000, dev)
del genome
g = mcx.Graph(31, 1, 1 << 24)
for N in (2, 4, 8, 16):
    segs, cap = g.superk_layout(N, batch.numel())
    recs = torch.empty((N, segs, cap, 2), dtype=torch.int64, device=dev)
    counts = torch.zeros((segs, N), dtype=torch.int64, device=dev)
    g.superk_bins_dev(batch, batch.numel(), N, recs, counts, cap); g.sync()
    k0 = g.device_stats().num_kmers_loaded
    counts.zero_()
    t0 = time.perf_counter()
    g.superk_bins_dev(batch, batch.numel(), N, recs, counts, cap); g.sync()
    dt = time.perf_counter() - t0
    occ = g.device_stats().num_kmers_loaded - k0
    per_owner = counts.sum(dim=0).double()
    print("N=%2d: %.2f records per 16 positions, %.2f B per occurrence (v2: 8.5), owner load max/mean %.3f, fill of the fullest segment %.0f %%, sender %.2f ms"
          % (N, float(per_owner.sum()) * 16 / batch.numel(), float(per_owner.sum()) * 16 / occ, float(per_owner.max() / per_owner.mean()),
             100.0 * float(counts.max()) / cap, dt * 1e3), flush=True)
    del recs, counts
