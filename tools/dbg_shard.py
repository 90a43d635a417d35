import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx
dev = torch.device("cuda", 0)
genome = bench.make_genome(50_000_000, dev, 42)
NR = int(os.environ.get("NR", "1000000"))
batches = [bench.make_batch(genome, NR, 1000 + i, dev) for i in range(3)]
ref = mcx.Graph(31, 1, 1 << 28)
for s in batches: ref.add_stream_dev(0, s, s.numel())
print("fused: nkmers", ref.nkmers, ref.device_stats().as_dict()); ref.close()
g = mcx.Graph(31, 1, 1 << 28)
g.configure("profile", 1)
ext = torch.cuda.ExternalStream(g.stream, device=dev)
cap = int(NR * 120 * 1.1) + 65536
sk = torch.empty((1, cap, 1), dtype=torch.int64, device=dev); se = torch.empty((1, cap), dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for s in batches:
    with torch.cuda.stream(ext): cnt.zero_()
    t0 = time.perf_counter()
    g.partition_stream_dev(s, s.numel(), 1, cap, sk, se, cnt); ext.synchronize()
    t1 = time.perf_counter()
    n = int(cnt.item())
    g.insert_tuples_dev(0, sk[0], se[0], n); ext.synchronize()
    t2 = time.perf_counter()
    print("n=%d partition %.1f ms insert %.1f ms" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter(); print("sharded: nkmers", g.nkmers, g.device_stats().as_dict(), "flush %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print(g.profile())
