#!/usr/bin/env python3
"""round 6: k = 65 .. 127 (three- and four-word keys, the fused insert kernel) at the benchmark's shape: 10 steps of 5 M x 150 bp
reads into 2^30 slots, device-resident stream; k = 31 and 63 through the same kernel (defer = 0) beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(bench.GENOME_PER_GPU, dev, seed=42)
batches = [bench.make_batch(genome, bench.BATCH_READS, seed=1000 + i, device=dev) for i in range(10)]
del genome
torch.cuda.empty_cache()
for k in (31, 63, 95, 127):
    g = mcx.Graph(k, 1, 1 << 30)
    g.configure("defer", 0)
    g.add_stream_dev(0, batches[0][:1024 * 151], 1024 * 151)
    g.sync(); g.reset(); g.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        g.add_stream_dev(0, b, b.numel())
    g.sync()
    dt = time.perf_counter() - t0
    st = g.device_stats()
    print("k = %3d (W = %d): %.1f ms, %.2f G k-mers/s, %d occurrences, %d distinct" % (k, g.W, dt * 1e3, st.num_kmers_loaded / dt / 1e9, st.num_kmers_loaded, g.nkmers), flush=True)
    g.close()
    torch.cuda.empty_cache()
