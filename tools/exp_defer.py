"""Experiment: direct vs deferred insert on the C2 workload; per-kernel time from the library's event spans."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

nsteps = int(os.environ.get("STEPS", "10"))
dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
batches = [bench.make_batch(genome, 5_000_000, 1000 + i, dev) for i in range(nsteps)]
del genome
torch.cuda.synchronize()
for cfg in sys.argv[1:] or ["defer=0", "defer=1", "defer=1,defer_tuples=4000000000"]:
    g = mcx.Graph(31, 1, 1 << 30)
    for kv in cfg.split(","):
        k, v = kv.split("=")
        g.configure(k, int(v))
    g.add_stream_dev(0, batches[0], batches[0].numel()); g.sync(); g.reset(); g.sync()
    g.configure("profile", 1)
    t0 = time.perf_counter()
    for s in batches:
        g.add_stream_dev(0, s, s.numel())
    g.sync()
    dt = time.perf_counter() - t0
    st = g.device_stats()
    print("%-40s %.1f ms total, %.2f G k-mers/s, distinct %d" % (cfg, dt * 1e3, st.num_kmers_loaded / dt / 1e9, st.num_kmers_novel))
    for name, (calls, ms) in g.profile().items():
        print("    %-18s calls %3d  total %8.2f ms  avg %7.3f ms" % (name, calls, ms, ms / calls))
    g.close()
    torch.cuda.empty_cache()
