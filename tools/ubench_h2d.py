"""Pinned host -> device copy rate on this box (what bounds the packed host entry): chunks of 48 MB, as staged."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (4, 48, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    n = max(4, 2048 // mb)
    t = time.perf_counter()
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("H2D %4d MB chunks: %.1f GB/s" % (mb, n * (mb << 20) / dt / 1e9), flush=True)
    h2 = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    t = time.perf_counter()
    for _ in range(n): h2.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("D2H %4d MB chunks: %.1f GB/s" % (mb, n * (mb << 20) / dt / 1e9), flush=True)
