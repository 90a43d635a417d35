#!/usr/bin/env python3
"""Experiment: does the VALU-bound k_stream_bin overlap with the HBM-bound k_tuples_bin when they
run on two streams?  Two handles: A k-merises batch n+1 into (owner, region) blocks while B splits
the blocks of batch n.  Prints serial vs overlapped time per batch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mccortex_amd as mcx
from mccortex_amd import shard

dev = torch.device("cuda", 0)
B, NB = 5_000_000, 6
genome = bench.make_genome(200_000_000, dev, 42)
batches = [bench.make_batch(genome, B, 1000 + i, dev) for i in range(NB)]
del genome
A = mcx.Graph(31, 1, 1 << 30)
A.configure("defer_tuples", 1 << 22)
Bg = mcx.Graph(31, 1, 1 << 30)
Bg.configure("defer_tuples", 8_000_000_000)
sa = torch.cuda.ExternalStream(A.stream, device=dev)
sb = torch.cuda.ExternalStream(Bg.stream, device=dev)
ntup = B * 120
segs, seg_cap, ov_cap = A.shard_layout(ntup)
blk = [shard.BlockExchange(1, segs, seg_cap, ov_cap, 1, dev) for _ in range(2)]
filled = [torch.cuda.Event() for _ in range(2)]
used = [torch.cuda.Event() for _ in range(2)]


def fill(i, b):
    with torch.cuda.stream(sa):
        blk[b].zero_counts()
    blk[b].fill(A, batches[i], batches[i].numel())
    filled[b].record(sa)


def consume(b):
    sb.wait_event(filled[b])
    blk[b].consume(Bg, 0, ntup)
    used[b].record(sb)


def run(overlap):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(NB):
        b = i % 2
        if i >= 2:
            sa.wait_event(used[b])
        fill(i, b)
        if not overlap:
            sa.synchronize()
        consume(b)
        if not overlap:
            sb.synchronize()
    sa.synchronize(); sb.synchronize()
    return (time.perf_counter() - t0) / NB * 1e3


for rep in range(2):
    for ov in (False, True):
        ms = run(ov)
        Bg.sync(); Bg.reset(); Bg.sync()
        print("overlap=%d  %.3f ms per batch (k_stream_bin + k_tuples_bin)" % (ov, ms), flush=True)
