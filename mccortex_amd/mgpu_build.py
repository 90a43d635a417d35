"""`build` over several GPUs of one node: one process per GPU, torch.distributed over RCCL.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m mccortex_amd.mgpu_build -k 31 -n 4G --sort --sample NA12878 --seq reads.fq out.ctx

SURVEY 8(e): every rank k-merises its share of every input file, one all-to-all per step moves
each occurrence (v3: each run of occurrences, as a super-k-mer record) to the rank that owns its
k-mer, the owners insert; the shards hold disjoint key sets, so the graph file is the header
(statistics summed over the ranks) followed by the shards' records -- merged by key for --sort.
The output is byte-identical to the single-GPU `mccortex<K> build` of the same command line.

Options are `build`'s (src/commands/ctx_build.c:13-77) where they make sense here:
-k -n -f -s/--sample -1/--seq -2/--seq2 -i/--seqi -S/--sort; -n is the capacity of the whole graph.
Not offered: --remove-pcr (needs the reads of a pair on one rank), -Q/-H, --graph, --intersect,
gzip input: the single-GPU command has them."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

STEP_BYTES = 256 << 20   # raw file bytes per rank per step (sizes the exchange buffers)
GROUP = 4                # steps resident in HBM at a time


def parse_size(s):
    """1024 2M 1G ... binary units (src/global/util.c:206-222)"""
    s = s.strip().upper().rstrip("B")
    mult = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30, "T": 1 << 40}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(s)


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="mgpu_build", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-k", "--kmer", type=int, required=True)
    ap.add_argument("-n", "--nkmers", default="4M")
    ap.add_argument("-f", "--force", action="store_true")
    ap.add_argument("-S", "--sort", action="store_true")
    ap.add_argument("-s", "--sample", action="append", default=[])
    ap.add_argument("-1", "--seq", action="append", default=[])
    ap.add_argument("-2", "--seq2", action="append", default=[])
    ap.add_argument("-i", "--seqi", action="append", default=[])
    ap.add_argument("--step-bytes", type=parse_size, default=STEP_BYTES)
    ap.add_argument("out")
    # sample / seq order matters (a --seq belongs to the --sample before it): walk argv ourselves
    tasks, names, colour = [], [], -1
    it = iter(range(len(argv)))
    for i in it:
        a = argv[i]
        if a in ("-s", "--sample"):
            colour += 1
            names.append(argv[i + 1]); next(it)
        elif a in ("-1", "--seq", "-i", "--seqi", "-2", "--seq2"):
            if colour < 0:
                raise SystemExit("Please give sample name first [-s,--sample <name>]")
            val = argv[i + 1]; next(it)
            if a in ("-2", "--seq2"):   # <in1>:<in2>: without --remove-pcr mates are loaded as two files (ctx_build.c:108-116)
                sep = ":" if ":" in val else ","
                if val.count(sep) != 1:
                    raise SystemExit("Expected -2 <in1>:<in2>")
                tasks.extend((f, colour) for f in val.split(sep))
            else:
                tasks.append((val, colour))
    args = ap.parse_args(argv)
    if not names:
        raise SystemExit("No inputs given")
    if not (args.kmer & 1):
        raise SystemExit("Invalid kmer-size (%d): requires odd number" % args.kmer)
    return args, tasks, names


def main(argv=None):
    args, tasks, names = parse_args(sys.argv[1:] if argv is None else argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.path.exists(args.out) and not args.force:   # every rank sees the same file system: all leave
        raise SystemExit("File already exists: %s" % args.out)
    import mccortex_amd as mcx
    from mccortex_amd import seqstream, shard

    # (MCX_DIST_BACKEND=gloo + MCX_DIST_ONE_DEVICE: the test transport of shard.py, several ranks on one GPU)
    local_rank = shard.local_device(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_PORT", "29541")
    shard.init_process_group(device, rank, world)

    k, ncols = args.kmer, len(names)
    # v3 (super-k-mer records, minimizer ownership: every rank holds an ordinary table) when k
    # allows it, else v2 (the table sharded by quotient-hash prefix); a rank owns ~1/N of the k-mers
    use_v3 = mcx.superk_supported(k) and os.environ.get("MCX_EXCHANGE", "v3") != "v2"
    per_rank = max(1024, int(parse_size(args.nkmers) * (1.15 if world > 1 else 1.0) / world))
    if use_v3:
        graph = mcx.Graph(k, ncols, per_rank, device=local_rank)
    else:
        graph = mcx.Graph(k, ncols, per_rank, device=local_rank, nparts=world, part=rank)   # (capacity is per shard)
    inserter = shard.ShardedInserter(graph, world, device, args.step_bytes + (1 << 16), use_v3)
    hdr = mcx.CtxHeader(k, ncols)
    hdr.names = list(names)
    empty = torch.zeros(16, dtype=torch.uint8, device=device)

    prev = graph.device_stats()
    for path, colour in tasks:
        fmt = seqstream.detect_format(path)
        mine = seqstream.plan_steps(path, fmt, rank, world, args.step_bytes)
        nsteps = torch.tensor([len(mine)], dtype=torch.int64, device=device)
        shard.all_reduce(nsteps, op=dist.ReduceOp.MAX)       # every rank runs the same number of exchanges
        nsteps = int(nsteps.item())
        for g0 in range(0, nsteps, GROUP):
            steps = []
            for i in range(g0, min(nsteps, g0 + GROUP)):
                if i < len(mine):
                    s = seqstream.to_stream(seqstream.load_bytes(path, mine[i][0], mine[i][1], device), fmt)
                    steps.append((s if s.numel() else empty, s.numel()))
                else:
                    steps.append((empty, 0))
            torch.cuda.synchronize()
            inserter.insert(colour, steps)
        # statistics of this file = the ranks' device counter deltas, summed (graph_info_update_stats
        # is fed once per input file, in task order: build_graph.c:294-298)
        cur = graph.device_stats()
        d = torch.tensor([cur.total_bases_loaded - prev.total_bases_loaded, cur.contigs_parsed - prev.contigs_parsed,
                          cur.num_kmers_loaded - prev.num_kmers_loaded], dtype=torch.int64, device=device)
        shard.all_reduce(d)
        prev = cur
        hdr.update_stats(colour, int(d[0].item()), int(d[1].item()))
        if rank == 0:
            print("[task] input: %s colour: %d  bases loaded: %d  contigs: %d  kmers: %d" % (path, colour, int(d[0]), int(d[1]), int(d[2])), file=sys.stderr)

    # every shard writes its records; rank 0 puts header and shards together
    part = "%s.part%d" % (args.out, rank)
    body = graph.export(bool(args.sort))
    with open(part, "wb") as f:
        f.write(body)
    nk = torch.tensor([graph.nkmers], dtype=torch.int64, device=device)
    shard.all_reduce(nk)
    dist.barrier()
    if rank == 0:
        rs = 8 * graph.W + 5 * ncols
        bodies = [open("%s.part%d" % (args.out, r), "rb").read() for r in range(world)]
        assert sum(len(b) for b in bodies) == int(nk.item()) * rs
        if args.sort and world > 1:
            # the shards are sorted and disjoint: one device sort of their concatenation merges them
            merged = mcx.sort_records(b"".join(bodies), k, ncols, device=local_rank)
        else:
            merged = b"".join(bodies)
        with open(args.out, "wb") as f:
            f.write(mcx.ctx_header_bytes(hdr))
            f.write(merged)
        for r in range(world):
            os.remove("%s.part%d" % (args.out, r))
        print("Dumped %d kmers in %d colour%s into: %s" % (int(nk.item()), ncols, "" if ncols == 1 else "s", args.out), file=sys.stderr)
    dist.barrier()
    graph.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
