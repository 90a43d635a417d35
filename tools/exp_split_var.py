#!/usr/bin/env python3
"""round 6: the split's time varies from graph to graph (19.5 .. 22.2 ms per 6 G occurrences at C2) and not from pass to
pass over one graph: with where which of its buffers lies?  One graph, one set of region bins; between passes the
sub-table bins are moved (MODE=l2: configure debug_realloc_l2), or a fresh graph is made (MODE=graph).  MCX_PLACE_BINS=n:
the library chooses among n placements itself; STEPS, WIN, MCX_CAP2_SLACK, BALLAST_GB as in profiles/r06_experiments.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(bench.GENOME_PER_GPU, dev, seed=42)
batches = [bench.make_batch(genome, bench.BATCH_READS, seed=1000 + i, device=dev) for i in range(int(os.environ.get("STEPS", "10")))]
del genome
torch.cuda.empty_cache()
mode = os.environ.get("MODE", "graph")


def mk():
    ballast = None
    if os.environ.get("BALLAST_GB"):  # push the graph's allocations up: memory below them is taken for the moment
        ballast = torch.empty(int(float(os.environ["BALLAST_GB"]) * (1 << 30)), dtype=torch.uint8, device=dev)
    g = mcx.Graph(31, 1, 1 << 30)
    g.configure("defer_tuples", int(os.environ.get("WIN", "8000000000")))
    g.configure("flush_overlap", 0)
    g.add_stream_dev(0, batches[0][:1024 * 151], 1024 * 151)
    g.sync(); g.reset(); g.sync()
    del ballast
    torch.cuda.empty_cache()
    return g


def one(g, tag):
    g.configure("profile", 1)
    for b in batches:
        g.add_stream_dev(0, b, b.numel())
    g.sync()
    prof = g.profile()
    print("%s %s: %s" % (mode, tag, "  ".join("%s %.2f" % (k, v[1]) for k, v in prof.items())), flush=True)
    g.reset(); g.sync()


keep = []
g = mk()
for it in range(int(os.environ.get("N", "8"))):
    one(g, "instance %d" % it)
    if mode == "graph":
        keep.append(torch.empty((it + 1) * (1 << 28), dtype=torch.uint8, device=dev))
        g.close()
        g = mk()
    elif mode == "l2":
        g.configure("debug_realloc_l2", (it + 1) << 28)
g.close()
