"""Experiment: the in-process multi-GPU table (mcx_graph_create_multi) with all shards on ONE GPU, C2 (10 x 5 M reads,
device-resident ASCII stream), 2 and 8 shards, exchange v3 / v2, copy kernel vs whole-block peer copies.  One process
per variant; prints G k-mers/s, the graph checksum (must be c72ff066d6a527ec) and the wall clock of the submission."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import bench, mccortex_amd as mcx
dev = torch.device("cuda", 0)
B = 5_000_000; N = 10
genome = bench.make_genome(200_000_000, dev, 42)
steps = [bench.make_batch(genome, B, 1000 + i, dev) for i in range(N)]
del genome; torch.cuda.empty_cache()
ns = int(os.environ.get("SHARDS", "8"))
g = mcx.Graph(31, 1, 1 << 30, devices=[0] * ns)
g.configure("defer_tuples", int(os.environ.get("DEFER", str(8_000_000_000 // max(ns, 2)))))   # (all shards share the ONE device's HBM here)
g.add_stream_dev(0, steps[0][:1024 * 151], 1024 * 151); g.sync(); g.reset(); g.sync()
res = []
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in steps: g.add_stream_dev(0, b, b.numel())
    t1 = time.perf_counter(); g.sync(); dt = time.perf_counter() - t0
    st = g.device_stats(); cs, nk = g.checksum()
    res.append("%%.1f G/s (%%.1f ms/step, submit %%.0f of %%.0f ms) cs %%016x" %% (st.num_kmers_loaded / dt / 1e9, 1e3 * dt / N, (t1 - t0) * 1e3, dt * 1e3, cs))
    g.reset(); g.sync()
print("; ".join(res))
''' % ROOT
variants = [dict(SHARDS="8"), dict(SHARDS="8", MCX_MULTI_COPY="memcpy"), dict(SHARDS="2"), dict(SHARDS="2", MCX_MULTI_COPY="memcpy"),
            dict(SHARDS="8", MCX_MULTI_EXCHANGE="v2"), dict(SHARDS="8", MCX_MULTI_EXCHANGE="v2", MCX_MULTI_COPY="memcpy")]
if len(sys.argv) > 1:
    variants = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else dict() for a in sys.argv[1:]]
for env in variants:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(env, p.stdout.decode().strip().splitlines()[-1:] or p.stderr.decode().strip().splitlines()[-3:], flush=True)
