"""Owner-side cost of exchange format v3 as a function of the number of owners the sender cut the
reads for: records get shorter (16 k-mers at 1 owner, ~6.6 at 8), the k-mers stay the same.
One GPU: the sender bins for `nparts` owners, ONE graph then consumes every owner's segments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mccortex_amd as mcx

dev = torch.device("cuda", 0)
genome = bench.make_genome(200_000_000, dev, 42)
batches = [bench.make_batch(genome, 5_000_000, 1000 + i, dev) for i in range(6)]
del genome
for nparts in (1, 8, 2):
    g = mcx.Graph(31, 1, 1 << 30)
    g.configure("defer_tuples", 8_000_000_000)
    segs, cap = g.superk_layout(nparts, batches[0].numel())
    recs = torch.zeros((nparts, segs, cap, 2), dtype=torch.int64, device=dev)
    fills = torch.zeros((segs, nparts), dtype=torch.int64, device=dev)
    g.add_stream_dev(0, batches[0][:151 * 1024], 151 * 1024); g.sync(); g.reset(); g.sync()
    g.configure("profile", 1)
    ext = torch.cuda.ExternalStream(g.stream, device=dev)
    nrec = 0
    for b in batches:
        fills.zero_(); torch.cuda.synchronize()
        g.superk_bins_dev(b, b.numel(), nparts, recs, fills, cap)
        ext.synchronize()   # (kernels run on the graph's stream, torch ops on torch's; g.sync() would flush)
        counts = fills.t().contiguous().clone()   # (a copy: fills is zeroed while the owner kernel may still read the counts)
        n = int(counts.sum())
        nrec += n
        g.add_superk_dev(0, recs, counts, nparts * segs, cap, b.numel())   # (an upper bound of the k-mers: one flush at the end)
    g.sync()
    st = g.device_stats()
    print("owners %2d: %.2f records per 16 positions, %.2f B per occurrence; ms per 600 M occurrences: %s" % (
        nparts, nrec * 16 / (len(batches) * batches[0].numel()), nrec * 16 / st.num_kmers_loaded,
        {k: (c, round(t / len(batches), 2)) for k, (c, t) in g.profile().items()}), flush=True)
    print("   checksum %016x nodes %d, occurrences counted by the senders %d" % (g.checksum() + (st.num_kmers_loaded,)), flush=True)
    g.close(); del recs; torch.cuda.empty_cache()
