import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
import mccortex_amd as mcx
from mccortex_amd import shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
genome = bench.make_genome(50_000_000, dev, 42)
NR = 2000000
s = bench.make_batch(genome, NR, 1000, dev)
g = mcx.Graph(31, 1, 1 << 28)
ext = torch.cuda.ExternalStream(g.stream, device=dev)
cap = int(NR * 120 * 1.1) + 65536
sk = torch.empty((1, cap, 1), dtype=torch.int64, device=dev); se = torch.empty((1, cap), dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
g.partition_stream_dev(s, s.numel(), 1, cap, sk, se, cnt); ext.synchronize()
n = int(cnt.item())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rk, re_, rc = shard.exchange(sk, se, cnt)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("exchange %.1f ms  n=%d rc=%s equal keys %s edges %s" % ((t1 - t0) * 1e3, n, rc, bool((rk[:, 0] == sk[0, :n, 0]).all()), bool((re_ == se[0, :n]).all())))
dist.destroy_process_group()
