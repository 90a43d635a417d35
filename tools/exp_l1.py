"""Experiment: time of the L1 kernel (k_stream_bin) alone for library variants (MCX_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx
dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
batches = [bench.make_batch(genome, 5_000_000, 1000 + i, dev) for i in range(3)]
g = mcx.Graph(31, 1, 1 << 30)
g.configure("defer_tuples", 4000000000)
g.add_stream_dev(0, batches[0], batches[0].numel()); g.reset(); g.sync()
g.configure("profile", 1)
for s in batches:
    g.add_stream_dev(0, s, s.numel())
torch.cuda.synchronize()
import ctypes
prof = g.profile()
print(os.environ.get("MCX_LIB", "default"), {k: round(v[1] / v[0], 3) for k, v in prof.items()})
sys.stdout.flush(); os._exit(0)
