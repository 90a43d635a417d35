import sys, os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
for n in (1_000_000, 50_000_000, 150_000_000, 268_435_455, 268_435_457, 300_000_000):
    src = torch.arange(n, dtype=torch.int64, device=dev).reshape(1, n, 1) * 7 + 3
    dst = torch.zeros((n, 1), dtype=torch.int64, device=dev)
    dist.all_to_all([dst[0:n]], [src[0, :n]])
    torch.cuda.synchronize()
    ok1 = bool((dst == src[0]).all())
    dst2 = torch.zeros((n, 1), dtype=torch.int64, device=dev)
    dist.all_to_all_single(dst2, src[0, :n].contiguous(), output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    ok2 = bool((dst2 == src[0]).all())
    nz = int((dst != src[0]).sum())
    print("n=%d (%.2f GB): list ok=%s (bad %d)  single ok=%s" % (n, n * 8 / 1e9, ok1, nz, ok2))
    del src, dst, dst2
dist.destroy_process_group()
