"""C4 (k = 63) and the 4-colour builds at full size, one line each with the graph checksum: the A/B
harness for kernel variants of those configs (MCX_LIB=build/variants/lib_x.so python tools/exp_c4c5.py [c4] [c5] [c5i])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

what = set(sys.argv[1:]) or {"c4", "c5", "c5i"}
dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
nsteps = int(os.environ.get("STEPS", "10"))
batches = [bench.make_batch(genome, 5_000_000, 1000 + i, dev) for i in range(nsteps)]
del genome
torch.cuda.synchronize()


def run(name, k, ncols, colours, slots=1 << 30, defer_tuples=6_000_000_000):
    g = mcx.Graph(k, ncols, slots)
    g.configure("defer_tuples", defer_tuples)
    g.add_stream_dev(0, batches[0][:151 * 1024], 151 * 1024); g.sync(); g.reset(); g.sync()
    g.configure("profile", 1)
    t0 = time.perf_counter()
    for i, s in enumerate(batches):
        g.add_stream_dev(colours[i], s, s.numel())
    g.sync()
    dt = time.perf_counter() - t0
    st = g.device_stats()
    cs, n = g.checksum()
    print("%-10s %-26s %.1f ms, %.2f G k-mers/s, %d distinct, cs %016x; %s" % (
        os.path.basename(os.environ.get("MCX_LIB", "default")), name, dt * 1e3, st.num_kmers_loaded / dt / 1e9, n, cs,
        {n_.replace("k_", ""): (c, round(t, 1)) for n_, (c, t) in g.profile().items()}), flush=True)
    g.close(); torch.cuda.empty_cache()


if "c2" in what:
    run("C2 k=31", 31, 1, [0] * nsteps, defer_tuples=8_000_000_000)
if "c4" in what:
    run("C4 k=63", 63, 1, [0] * nsteps, defer_tuples=int(os.environ.get("C4_DEFER", "5000000000")))
if "c5" in what:
    run("C5-like 4 colours blocks", 31, 4, [min(3, 4 * i // nsteps) for i in range(nsteps)], defer_tuples=int(os.environ.get("C5_DEFER", "6000000000")))
if "c5i" in what:
    run("C5-like 4 colours interleaved", 31, 4, [i % 4 for i in range(nsteps)], defer_tuples=int(os.environ.get("C5_DEFER", "6000000000")))
