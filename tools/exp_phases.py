#!/usr/bin/env python3
"""Per-phase time of the three build kernels (variant library built with -DMCX_PHASES, see csrc/mcx_defer.h):
   tools/variants.sh phases:"-DMCX_PHASES" ; MCX_LIB=$PWD/build/variants/lib_phases.so python tools/exp_phases.py
One C2 build (10 steps, one flush, flush overlap off) and one C4 build (k = 63); prints, per kernel, the share of
thread 0's wall time spent in each phase and the mean time per tile / sub-table visit."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mccortex_amd as mcx

NAMES = {
    0: ("k_stream_bin", ["stage tile + zero counters + barrier", "masks + 16 positions (k-mer, canonical, hash, histogram atomic)",
                         "bin_reserve (scan, 3 barriers, reservations issued)", "arrival -> sorted position, bin_commit",
                         "round 0: place + barrier + write-out", "round 1: place + barrier + write-out", "loop top", "tiles"]),
    1: ("k_tuples_bin", ["zero counters + 2 barriers", "tuple loads + sub_hash + histogram atomic", "bin_reserve", "rank atomics + bin_commit",
                         "round 0: place + write-out", "round 1: place + write-out", "chunk index arithmetic / skipped chunks", "tiles"]),
    2: ("k_lds_insert", ["fill count + next_bin + barrier", "slice registers -> LDS + barrier", "tuple batches: load / unpack / lds_try",
                         "barrier after the batches", "queue (lds_apply) + barrier", "slice LDS -> HBM", "loop top", "sub-table visits"]),
}


if os.environ.get("MCX_STREAM_FC", "0") not in ("", "0"):  # k_stream_fc (mcx_streamfc.h) in place of k_stream_bin
    NAMES[0] = ("k_stream_fc", ["masks + 16 positions + 16 segment stores", "pool append", "barrier after k-merising",
                                "reservations issued + next tile staged + pool drained", "wait for reservations + copy-out",
                                "end barrier", "loop top", "tiles"])


def report(L, title):
    buf = (C.c_uint64 * 24)()
    assert L.mcx_debug_phases(buf, 1) == 0
    print("== " + title)
    for kid, (kname, phases) in NAMES.items():
        v = [int(buf[kid * 8 + i]) for i in range(8)]
        tot = sum(v[:7])
        if not tot:
            continue
        n = max(1, v[7])
        print("%s: %d %s, %.2f us each (thread 0 of every block, 100 MHz clock)" % (kname, v[7], phases[7], tot / n / 100.0))
        for i in range(7):
            print("    %5.1f %%  %7.2f us  %s" % (100.0 * v[i] / tot, v[i] / n / 100.0, phases[i]))


def main():
    dev = torch.device("cuda", 0)
    L = mcx.lib()
    L.mcx_debug_phases.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    if os.environ.get("PHASES_CFG") == "hashtest":  # the reference's hashtest: 800 M integer keys into 2^30 slots
        n = 800_000_000
        keys = torch.arange(n, dtype=torch.int64, device=dev)
        edges = torch.zeros(n, dtype=torch.uint8, device=dev)
        g = mcx.Graph(31, 1, 1 << 30)
        g.configure("flush_overlap", 0)
        g.insert_tuples_dev(0, keys[:65536], edges[:65536], 65536)
        g.sync(); g.reset(); g.sync()
        L.mcx_debug_phases(None, 1)
        g.configure("profile", 1)
        for lo in range(0, n, 100_000_000):
            g.insert_tuples_dev(0, keys[lo:lo + 100_000_000], edges[lo:lo + 100_000_000], 100_000_000)
        g.sync()
        prof = g.profile()
        print("nkmers", g.nkmers)
        g.close()
        report(L, "hashtest: %s" % "  ".join("%s %.2f ms (%d)" % (a, t, c) for a, (c, t) in prof.items()))
        return
    stress = os.environ.get("PHASES_CFG") == "stress"  # C2-stress: iid reads (every k-mer novel), 2^33 slots
    if stress:
        batches = [bench.make_batch_iid(bench.BATCH_READS, seed=7000 + i, device=dev) for i in range(10)]
    else:
        genome = bench.make_genome(bench.GENOME_PER_GPU, dev, seed=42)
        batches = [bench.make_batch(genome, bench.BATCH_READS, seed=1000 + i, device=dev) for i in range(10)]
        del genome
    torch.cuda.empty_cache()
    for k, defer in (((31, 6_300_000_000),) if stress else ((31, 8_000_000_000), (63, 5_000_000_000))):
        g = mcx.Graph(k, 1, (1 << 33) if stress else (1 << 30))
        g.configure("defer_tuples", defer)
        g.configure("flush_overlap", 0)
        g.add_stream_dev(0, batches[0][:1024 * 151], 1024 * 151)
        g.sync(); g.reset(); g.sync()
        L.mcx_debug_phases(None, 1)
        g.configure("profile", 1)
        for b in batches:
            g.add_stream_dev(0, b, b.numel())
        g.sync()
        prof = g.profile()
        cs, n = g.checksum()
        g.close()
        torch.cuda.empty_cache()
        report(L, "k = %d: %s  checksum %016x" % (k, "  ".join("%s %.2f ms" % (a, t) for a, (c, t) in prof.items()), cs))


if __name__ == "__main__":
    main()
