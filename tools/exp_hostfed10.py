"""Experiment: the bench's host_fed leg (10 x 5 M reads from pinned host memory through
mcx_graph_add_reads, flush inside the clock) under staging variants; one process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench, mccortex_amd as mcx
dev = torch.device("cuda", 0)
B = 5_000_000; N = 10
genome = bench.make_genome(200_000_000, dev, 42)
hb = []
for i in range(N):
    b = bench.make_batch(genome, B, 1000 + i, dev)
    t = torch.empty((B, 150), dtype=torch.uint8).pin_memory()
    t.copy_(b.reshape(B, 151)[:, :150]); hb.append(t.numpy().reshape(-1)); del b
del genome; torch.cuda.empty_cache()
offs = np.arange(B + 1, dtype=np.uint64) * 150
g = mcx.Graph(31, 1, 1 << 30)
g.add_reads(0, hb[0][:150000], offs[:1001]); g.sync(); g.reset(); g.sync()
res = []
for rep in range(2):
    t0 = time.perf_counter()
    for h in hb: g.add_reads(0, h, offs)
    t1 = time.perf_counter(); g.sync(); dt = time.perf_counter() - t0
    res.append("%%.1f G/s (submit %%.0f of %%.0f ms)" %% (N * B * 120 / dt / 1e9, (t1 - t0) * 1e3, dt * 1e3))
    g.reset(); g.sync()
print("; ".join(res))
''' % ROOT
variants = [dict(), dict(MCX_IDLE_FLUSH="0"), dict(MCX_STAGE_THREADS="24")]
if len(sys.argv) > 1:  # variants from the command line: "A=1,B=2" per argument ("-" = the defaults)
    variants = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else dict() for a in sys.argv[1:]]
for env in variants:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    print(env, p.stdout.decode().strip().splitlines()[-1:], flush=True)
