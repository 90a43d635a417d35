"""Experiment: speed of the kmerize+partition kernel (MODE 1) vs number of bins."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
s = bench.make_batch(genome, 5_000_000, 1, dev)
g = mcx.Graph(31, 1, 1 << 20)
ext = torch.cuda.ExternalStream(g.stream, device=dev)
for nparts in (1, 8, 64):
    cap = int(600e6 / nparts * 1.1) + 65536
    keys = torch.empty((nparts, cap, 1), dtype=torch.int64, device=dev)
    edges = torch.empty((nparts, cap), dtype=torch.uint8, device=dev)
    counts = torch.zeros(nparts, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for it in range(3):
        counts.zero_(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ext)
        g.partition_stream_dev(s, s.numel(), nparts, cap, keys, edges, counts)
        b.record(ext)
        g.sync()
        n = int(counts.sum().item())
    print("nparts=%d: %.2f ms for %d tuples -> %.1f G tuples/s, %.1f GB/s written" %
          (nparts, a.elapsed_time(b), n, n / a.elapsed_time(b) / 1e6, n * 9 / a.elapsed_time(b) / 1e6))
    del keys, edges
