#!/usr/bin/env python3
"""Experiment: effective bandwidth of produce->consume ping-pong through buffers of various sizes
(does a ~100 MB intermediate stay in the 256 MiB Infinity Cache between two kernels?)."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (32, 64, 96, 128, 192, 256, 512, 2048, 8192):
    n = mb * (1 << 20) // 8
    a = torch.zeros(n, dtype=torch.int64, device=dev)
    b = torch.zeros(n, dtype=torch.int64, device=dev)
    iters = max(4, 16384 // mb)
    for _ in range(2):
        b.copy_(a); a.copy_(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        b.copy_(a)
        a.copy_(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%5d MB buffers: %.0f GB/s (read+write), %.1f us per copy" % (mb, 4 * n * 8 * iters / dt / 1e9, dt / iters / 2 * 1e6), flush=True)
    del a, b
