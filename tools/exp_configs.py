"""Other BASELINE configs at full size (C4: k=63; C5-like: 4 colours) and the PCIe-inclusive host entry."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
nsteps = 10
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
    print("%-28s %.1f ms, %.2f G k-mers/s, %d occurrences, %d distinct; %s" % (
        name, dt * 1e3, st.num_kmers_loaded / dt / 1e9, st.num_kmers_loaded, st.num_kmers_novel,
        {n: round(t, 1) for n, (c, t) in g.profile().items()}), flush=True)
    g.close(); torch.cuda.empty_cache()

run("C2 k=31 1 colour", 31, 1, [0] * nsteps, defer_tuples=8_000_000_000)
run("C4 k=63 1 colour", 63, 1, [0] * nsteps, defer_tuples=5_000_000_000)
run("C5-like k=31 4 colours", 31, 4, [0, 0, 0, 1, 1, 1, 2, 2, 3, 3])
run("k=21 1 colour", 21, 1, [0] * nsteps, defer_tuples=8_000_000_000)

# PCIe-inclusive host entry: 5M reads x 150 bp in host memory (numpy), mcx_graph_add_reads
s = batches[0].reshape(-1, 151)[:, :150].contiguous().cpu().numpy().reshape(-1)
offs = np.arange(5_000_001, dtype=np.uint64) * 150
g = mcx.Graph(31, 1, 1 << 30)
g.add_reads(0, s[:150 * 1000], offs[:1001]); g.sync(); g.reset(); g.sync()
t0 = time.perf_counter()
g.add_reads(0, s, offs)
t1 = time.perf_counter()
g.sync()
t2 = time.perf_counter()
st = g.device_stats()
print("host entry (pinned staging, 1 host thread): submit %.1f ms, total %.1f ms -> %.2f G k-mers/s, %.2f GB/s of bases" % (
    (t1 - t0) * 1e3, (t2 - t0) * 1e3, st.num_kmers_loaded / (t2 - t0) / 1e9, len(s) / (t2 - t0) / 1e9), flush=True)
