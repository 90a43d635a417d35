import torch, time
dev = torch.device("cuda", 0)
for gb in (1, 4, 16):
    n = gb << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_(); b.zero_(); torch.cuda.synchronize()
    for _ in range(2): b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("D2D copy %2d GiB: %.2f TB/s (read + write)" % (gb, 2 * n / dt / 1e12), flush=True)
    t0 = time.perf_counter()
    for _ in range(5): a.zero_()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("memset   %2d GiB: %.2f TB/s" % (gb, n / dt / 1e12), flush=True)
    t0 = time.perf_counter()
    for _ in range(5): s = a.view(torch.int64).sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("read-sum %2d GiB: %.2f TB/s" % (gb, n / dt / 1e12), flush=True)
    del a, b
