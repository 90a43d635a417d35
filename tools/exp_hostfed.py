"""Experiment: host-fed entry (mcx_graph_add_reads from pinned host memory) under staging variants.
Every configuration runs in its own process (the knobs are read once): MCX_PACKED, MCX_STAGE_THREADS, MCX_STAGE_BYTES."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench, mccortex_amd as mcx
dev = torch.device("cuda", 0)
B = 5_000_000
genome = bench.make_genome(200_000_000, dev, 42)
hb = []
for i in range(2):
    b = bench.make_batch(genome, B, 1000 + i, dev)
    t = torch.empty((B, 150), dtype=torch.uint8).pin_memory()
    t.copy_(b.reshape(B, 151)[:, :150]); hb.append(t.numpy().reshape(-1)); del b
del genome; torch.cuda.empty_cache()
offs = np.arange(B + 1, dtype=np.uint64) * 150
g = mcx.Graph(31, 1, 1 << 30)
g.add_reads(0, hb[0][:150000], offs[:1001]); g.sync(); g.reset(); g.sync()
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for h in hb: g.add_reads(0, h, offs)
    t1 = time.perf_counter(); g.sync(); dt = time.perf_counter() - t0
    st = g.device_stats(); best = max(best, 2 * B * 120 / dt)
    g.reset(); g.sync()
print("%%.2f G k-mers/s (submit %%.1f ms of %%.1f ms)" %% (best / 1e9, (t1 - t0) * 1e3, dt * 1e3))
''' % ROOT
for env in [dict(MCX_PACKED="0", MCX_STAGE_THREADS="4"), dict(MCX_PACKED="0", MCX_STAGE_THREADS="16"),
            dict(MCX_PACKED="1", MCX_STAGE_THREADS="4"), dict(MCX_PACKED="1", MCX_STAGE_THREADS="16"),
            dict(MCX_PACKED="1", MCX_STAGE_THREADS="32"), dict(MCX_PACKED="1", MCX_STAGE_THREADS="16", MCX_STAGE_BYTES=str(128 << 20)),
            dict(MCX_PACKED="1", MCX_STAGE_THREADS="32", MCX_STAGE_BYTES=str(128 << 20)), dict(MCX_PACKED="1", MCX_STAGE_THREADS="16", MCX_NO_AVX2="1")]:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    print(env, p.stdout.decode().strip().splitlines()[-1:], flush=True)
