"""Experiment: how much of k_stream_bin (instruction-issue bound) hides under a flush (k_tuples_bin +
k_lds_insert, HBM bound) when they run on two streams with grids small enough to share the CUs.
Two handles on one GPU: X k-merises fresh batches, Y flushes 10 buffered steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mccortex_amd as mcx
dev = torch.device("cuda", 0)
N = 10
genome = bench.make_genome(200_000_000, dev, 42)
batches = [bench.make_batch(genome, 5_000_000, 1000 + i, dev) for i in range(N)]
del genome
packed = bench.pack_batches(mcx, batches)

def mk(defer):
    g = mcx.Graph(31, 1, 1 << 30)
    g.configure("defer_tuples", defer)
    g.add_packed_dev(0, packed[0][0][:4096], packed[0][1][:4096], 65536); g.sync(); g.reset(); g.sync()
    return g

def fill(g):
    for p in packed: g.add_packed_dev(0, *p)

def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

X, Y = mk(8_000_000_000), mk(8_000_000_000)
sx = torch.cuda.ExternalStream(X.stream, device=dev); sy = torch.cuda.ExternalStream(Y.stream, device=dev)
for gs, gp, gi in [(0, 0, 0), (1024, 2048, 1024), (512, 1024, 512), (768, 1024, 512), (512, 2048, 512), (1024, 1024, 512)]:
    for g in (X, Y):
        g.configure("grid_stream", gs); g.configure("grid_split", gp); g.configure("grid_insert", gi)
    X.reset(); Y.reset(); X.sync(); Y.sync()
    a = t(lambda: fill(X))                     # k-merise alone (into X's bins)
    fill(Y); torch.cuda.synchronize()
    b = t(lambda: Y.sync())                    # flush alone
    X.reset(); X.sync(); fill(Y); torch.cuda.synchronize()
    def both():
        fill(X)                                # async on X's stream
        Y.sync()                               # flush on Y's stream meanwhile (host blocks here)
    c = t(both)
    print("grids stream/split/insert %4d/%4d/%4d: k-merise %.1f ms, flush %.1f ms, serial %.1f, concurrent %.1f ms" % (gs, gp, gi, a, b, a + b, c), flush=True)
    X.sync()
