"""Hash-prefix sharded build over torch.distributed (RCCL on the GPU box, gloo in CPU tests).

The data path is exchange-only: every rank bins its per-occurrence tuples by owner and one
all-to-all moves each bin to the rank that owns it.  `exchange()` is backend-agnostic host logic;
the device kernels are reached through the Graph handle (mccortex_amd/graph.py)."""
import os

import numpy as np
import torch
import torch.distributed as dist

MAX_ROUND = 1 << 24  # tuples per peer per all-to-all round (128 MiB of keys at W=1)


# ---- transport ---------------------------------------------------------------------------------
# The product transport is RCCL ("nccl" backend: device buffers straight onto xGMI).  RCCL refuses two
# ranks on one device, so a box with ONE GPU can never run rank 1 -- MCX_DIST_BACKEND=gloo is the
# test-only transport for that box: every collective below stages device tensors through host tensors
# (device -> host copy, gloo collective over TCP loopback, host -> device copy, all in the order of
# torch's current stream), and MCX_DIST_ONE_DEVICE=<d> puts every rank on device d.  Everything else --
# sender / owner kernels, buffer rotation, events, rank-dependent slicing -- is the code RCCL runs with.

def dist_backend():
    return os.environ.get("MCX_DIST_BACKEND", "nccl")


def local_device(local_rank):
    """device index of this rank: LOCAL_RANK, or MCX_DIST_ONE_DEVICE for all ranks (test transport only)"""
    one = os.environ.get("MCX_DIST_ONE_DEVICE")
    if one is not None:
        if dist_backend() == "nccl":
            raise SystemExit("MCX_DIST_ONE_DEVICE needs MCX_DIST_BACKEND=gloo (RCCL refuses two ranks on one GPU)")
        return int(one)
    return local_rank


def init_process_group(device, rank, world):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dist_backend() == "nccl":
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
    else:
        dist.init_process_group(dist_backend(), rank=rank, world_size=world)


def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) != "nccl"


def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def barrier(group=None):
    dist.barrier(group=group)


def all_to_all_single(dst, src, output_split_sizes=None, input_split_sizes=None, group=None):
    if _staged(src, group):
        hs, hd = src.contiguous().cpu(), torch.empty(dst.shape, dtype=dst.dtype)
        dist.all_to_all_single(hd, hs, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        dst.copy_(hd)
    else:
        dist.all_to_all_single(dst, src, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)


def exchange(send_keys, send_edges, counts, group=None):
    """send_keys [world, cap, W] int64, send_edges [world, cap] uint8, counts [world] int64 (same
    device).  Returns (recv_keys [n, W], recv_edges [n], recv_counts list) for this rank."""
    world = dist.get_world_size(group)
    recv_counts = torch.empty_like(counts)
    all_to_all_single(recv_counts, counts, group=group)
    sc, rc = counts.tolist(), recv_counts.tolist()
    ro = np.concatenate([[0], np.cumsum(rc)]).astype(np.int64)
    W = send_keys.shape[2]
    n_in = int(ro[-1])
    recv_keys = torch.empty((n_in, W), dtype=send_keys.dtype, device=send_keys.device)
    recv_edges = torch.empty((n_in,), dtype=send_edges.dtype, device=send_edges.device)
    # Messages are cut into rounds of at most MAX_ROUND tuples per peer: RCCL (2.26, ROCm 7.0)
    # was observed to deliver only part of an all-to-all message larger than ~1 GB
    # (tools/dbg_shard3.py), and bounded rounds also bound the staging RCCL needs.
    # every rank must run the same number of rounds: agree on the largest per-peer message
    gmax = torch.tensor([max(max(sc), max(rc), 1)], dtype=torch.int64, device=counts.device)
    all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    for r0 in range(0, int(gmax.item()), MAX_ROUND):
        s_lo = [min(c, r0) for c in sc]
        s_hi = [min(c, r0 + MAX_ROUND) for c in sc]
        r_lo = [min(c, r0) for c in rc]
        r_hi = [min(c, r0 + MAX_ROUND) for c in rc]
        if dist.get_backend(group) == "nccl":
            # RCCL takes per-peer tensor lists: the filled bin prefixes go out in place (no packing
            # copy); every pair of GPUs is one xGMI hop, so this is a direct all-to-all, not a ring
            dist.all_to_all([recv_keys[ro[p] + r_lo[p]:ro[p] + r_hi[p]] for p in range(world)],
                            [send_keys[p, s_lo[p]:s_hi[p]] for p in range(world)], group=group)
            dist.all_to_all([recv_edges[ro[p] + r_lo[p]:ro[p] + r_hi[p]] for p in range(world)],
                            [send_edges[p, s_lo[p]:s_hi[p]] for p in range(world)], group=group)
        else:
            # gloo (CPU tests) has no list all_to_all: pack the round and use split sizes
            pk = torch.cat([send_keys[p, s_lo[p]:s_hi[p]] for p in range(world)])
            pe = torch.cat([send_edges[p, s_lo[p]:s_hi[p]] for p in range(world)])
            ins = [s_hi[p] - s_lo[p] for p in range(world)]
            outs = [r_hi[p] - r_lo[p] for p in range(world)]
            tk = torch.empty((sum(outs), W), dtype=send_keys.dtype, device=send_keys.device)
            te = torch.empty((sum(outs),), dtype=send_edges.dtype, device=send_edges.device)
            all_to_all_single(tk, pk, output_split_sizes=outs, input_split_sizes=ins, group=group)
            all_to_all_single(te, pe, output_split_sizes=outs, input_split_sizes=ins, group=group)
            o = 0
            for p in range(world):
                recv_keys[ro[p] + r_lo[p]:ro[p] + r_hi[p]] = tk[o:o + outs[p]]
                recv_edges[ro[p] + r_lo[p]:ro[p] + r_hi[p]] = te[o:o + outs[p]]
                o += outs[p]
    return recv_keys, recv_edges, rc


MAX_BLOCK_BYTES = 512 << 20  # per-peer message of one all-to-all round of the block exchange


def _a2a_rows(dst, src, group):
    """dst[p] <- peer p's src[my rank], for [world, rows, ...] tensors with contiguous rows, in
    rounds of at most MAX_BLOCK_BYTES per peer (same RCCL message-size limit as in exchange();
    the rounds are static because the blocks have a fixed size)."""
    world = dist.get_world_size(group)
    rows = src.shape[1]
    row_bytes = max(1, src[0, 0].numel() * src.element_size()) if rows else 1
    step = max(1, MAX_BLOCK_BYTES // row_bytes)
    nccl = dist.get_backend(group) == "nccl"
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        if r0 == 0 and r1 == rows:
            all_to_all_single(dst, src, group=group)
        elif nccl:
            dist.all_to_all([dst[p, r0:r1] for p in range(world)], [src[p, r0:r1] for p in range(world)], group=group)
        else:
            tmp = torch.empty_like(src[:, r0:r1].contiguous())
            all_to_all_single(tmp, src[:, r0:r1].contiguous(), group=group)
            dst[:, r0:r1] = tmp


class BlockExchange:
    """Exchange format v2 (mcx_graph_shard_bins_dev / mcx_graph_add_segments_dev): every rank holds
    keys[world][segs][seg_cap][W] packed tuples binned by (owner, region), fills counts[world][segs],
    and per-owner overflow bins of full tuples.  All blocks have a fixed size, so the all-to-all
    needs no count round trip and no host synchronisation; padding costs the slack of
    mcx_graph_shard_layout (~6-12 %) in link bytes."""

    def __init__(self, world, segs, seg_cap, ov_cap, W, device):
        self.world, self.segs, self.seg_cap, self.ov_cap, self.W = world, segs, seg_cap, ov_cap, W
        i64 = dict(dtype=torch.int64, device=device)
        self.keys = torch.empty((world, segs, seg_cap, W), **i64)
        self.counts = torch.zeros((world, segs), **i64)
        self.ov_keys = torch.empty((world, ov_cap, W), **i64)
        self.ov_edges = torch.empty((world, ov_cap), dtype=torch.uint8, device=device)
        self.ov_counts = torch.zeros((world,), **i64)

    def zero_counts(self):
        self.counts.zero_()
        self.ov_counts.zero_()

    def fill(self, graph, d_stream, nbytes):
        """sender: k-merise a resident stream into this block set (counts must be zero)"""
        graph.shard_bins_dev(d_stream, nbytes, self.keys, self.counts, self.seg_cap, self.ov_keys, self.ov_edges,
                             self.ov_counts, self.ov_cap)

    def exchange_into(self, recv, group=None):
        """recv.X[p] <- rank p's X[my rank] for every buffer (collectives on the current stream)"""
        _a2a_rows(recv.counts.view(self.world, 1, self.segs), self.counts.view(self.world, 1, self.segs), group)
        _a2a_rows(recv.ov_counts.view(self.world, 1, 1), self.ov_counts.view(self.world, 1, 1), group)
        _a2a_rows(recv.keys, self.keys, group)
        _a2a_rows(recv.ov_keys.view(self.world, 1, -1), self.ov_keys.view(self.world, 1, -1), group)
        _a2a_rows(recv.ov_edges.view(self.world, 1, -1), self.ov_edges.view(self.world, 1, -1), group)
        # whole blocks travel whatever their fill: bytes this rank puts on the links (to the other ranks)
        per_peer = sum(t[0].numel() * t.element_size() for t in (self.keys, self.counts, self.ov_keys, self.ov_edges, self.ov_counts))
        return None, per_peer * (self.world - 1)

    def consume(self, graph, colour, ntuples):
        """owner: hand received blocks (this object is a receive set) to the graph"""
        graph.add_segments_dev(colour, self.keys, self.counts, self.world * self.segs, self.seg_cap, ntuples)
        graph.insert_tuple_segments_dev(colour, self.ov_keys, self.ov_edges, self.ov_counts, self.world, self.ov_cap)

    def overflowed(self):
        """sender-side check (host sync): did an overflow bin itself overflow? (mcx_graph_sync on the
        sending graph reports the same condition as MCX_ERR_FULL)"""
        return bool((self.ov_counts > self.ov_cap).any().item())


class SuperkExchange:
    """Exchange format v3 (mcx_graph_superk_bins_dev / mcx_graph_add_superk_dev): per-owner bins of
    16-byte (two-word keys: 32-byte) super-k-mer records, `segs` replica segments per owner.  A send set keeps its fills
    replica-major (fills[segs][world], what the kernel wants); a receive set per source
    (counts[world][segs], one fill per received segment).
    Only the filled part of every segment travels: the fills are exchanged first (one small
    all-to-all and one host read), then one all-to-all per replica segment moves the records."""

    def __init__(self, world, segs, seg_cap, device, rec_words=2):
        self.world, self.segs, self.seg_cap, self.rec_words = world, segs, seg_cap, rec_words
        self.recs = torch.empty((world, segs, seg_cap, rec_words), dtype=torch.int64, device=device)
        self.fills = torch.zeros((segs, world), dtype=torch.int64, device=device)    # sender side
        self.counts = torch.zeros((world, segs), dtype=torch.int64, device=device)   # receiver side

    def zero_counts(self):
        self.fills.zero_()

    def fill(self, graph, d_stream, nbytes):
        graph.superk_bins_dev(d_stream, nbytes, self.world, self.recs, self.fills, self.seg_cap)

    def exchange_into(self, recv, group=None):
        """recv.recs[p][s][:n] <- rank p's recs[my rank][s][:n]; returns the records received"""
        world = self.world
        mine = self.fills.t().contiguous()                           # [owner][segment]
        all_to_all_single(recv.counts, mine, group=group)
        sc = torch.clamp(mine, max=self.seg_cap).tolist()             # host read: this step's fills
        rc = torch.clamp(recv.counts, max=self.seg_cap).tolist()
        nccl = dist.get_backend(group) == "nccl"
        for s_ in range(self.segs):
            if nccl:
                dist.all_to_all([recv.recs[p, s_, :rc[p][s_]] for p in range(world)],
                                [self.recs[p, s_, :sc[p][s_]] for p in range(world)], group=group)
            else:
                ins = [sc[p][s_] for p in range(world)]
                outs = [rc[p][s_] for p in range(world)]
                pk = torch.cat([self.recs[p, s_, :ins[p]] for p in range(world)])
                tk = torch.empty((sum(outs), self.rec_words), dtype=self.recs.dtype, device=self.recs.device)
                all_to_all_single(tk, pk, output_split_sizes=outs, input_split_sizes=ins, group=group)
                o = 0
                for p in range(world):
                    recv.recs[p, s_, :outs[p]] = tk[o:o + outs[p]]
                    o += outs[p]
        me = dist.get_rank(group)
        sent = sum(sum(sc[p]) for p in range(world) if p != me) * self.rec_words * 8 + (world - 1) * self.segs * 8
        return sum(sum(r) for r in rc), sent

    def consume(self, graph, colour, nrecords):
        """owner: k-merise the received segments (this object is a receive set)"""
        graph.add_superk_dev(colour, self.recs, self.counts, self.world * self.segs, self.seg_cap, nrecords * 16)  # <= 16 k-mers per record

    def overflowed(self):
        return bool((self.fills > self.seg_cap).any().item())


class ShardedInserter:
    """The N > 1 build step, shared by bench.py and the multi-GPU build tool (mgpu_build.py):
    every rank cuts its own reads into per-owner bins (sender kernel, the graph's stream), one
    all-to-all per step moves each bin to its owner (RCCL, torch's current stream) and the owner
    inserts what it received (the graph's stream).  Send and receive sets are double buffered: the
    sender kernel of step n+1 overlaps with the all-to-all of step n and with the owner-side
    kernels of step n-1.  Nothing here reads the graph, so nothing flushes it.

    Exchange format v3 (super-k-mer records, minimizer ownership, ordinary per-rank tables) when
    `use_v3`; else v2 (packed tuples binned by (owner, region), the table sharded by quotient-hash
    prefix: the graph must have been created with nparts = world, part = rank)."""

    def __init__(self, graph, world, device, max_stream_bytes, use_v3, group=None, max_tuples=None):
        self.graph, self.world, self.use_v3, self.group = graph, world, use_v3, group
        self.max_stream_bytes = int(max_stream_bytes)
        # k-mer occurrences one step can hold (<= its bytes): sizes the v2 blocks
        self.max_tuples = int(max_tuples) if max_tuples is not None else self.max_stream_bytes
        self.ext = torch.cuda.ExternalStream(graph.stream, device=device)
        if use_v3:
            # the sender only computes minimizers and cuts the reads into per-owner records (16
            # bytes per run of <= 16 k-mers: ~2.3 B per occurrence on the links instead of 8.5);
            # the owner k-merises what it receives.  Only filled parts travel, which costs one
            # host read of the fills per step.
            segs, seg_cap = graph.superk_layout(world, self.max_stream_bytes)
            mk = lambda: SuperkExchange(world, segs, seg_cap, device, rec_words=2 * graph.W)
        else:
            # every owner's block has a fixed size: one all-to-all per buffer, no count round trip
            segs, seg_cap, ov_cap = graph.shard_layout(self.max_tuples)
            mk = lambda: BlockExchange(world, segs, seg_cap, ov_cap, graph.W, device)
        self.send = [mk() for _ in range(2)]
        self.recv = [mk() for _ in range(2)]
        self.filled = [torch.cuda.Event() for _ in range(2)]     # send[b] k-merised             (ext)
        self.sent = [torch.cuda.Event() for _ in range(2)]       # send[b] -> recv[b] exchanged   (torch)
        self.consumed = [torch.cuda.Event() for _ in range(2)]   # recv[b] split by the owner     (ext)
        self.used = [False, False]                               # send[b] has been exchanged before
        # diagnostics (bench.py --gpus N prints them per rank): the exchange's span on torch's stream per
        # step (HIP events, read after the closing synchronize) and the bytes this rank put on the links
        self.reset_stats()

    def reset_stats(self):
        self.stats = {"steps": 0, "exchange_ms": 0.0, "link_bytes_sent": 0, "stream_bytes": 0}

    def _partition(self, stream, nbytes, buf):
        if nbytes > self.max_stream_bytes:
            raise ValueError("a step of %d bytes exceeds the %d the exchange buffers were sized for" % (nbytes, self.max_stream_bytes))
        with torch.cuda.stream(self.ext):
            if self.used[buf]:
                self.ext.wait_event(self.sent[buf])   # the previous exchange out of this send set is over
            self.send[buf].zero_counts()
        self.send[buf].fill(self.graph, stream, nbytes)
        self.filled[buf].record(self.ext)

    def insert(self, colour, steps):
        """steps: list of (device byte stream, nbytes), reads separated by a non-ACGT byte; EVERY
        rank must pass the same number of steps (an empty one is (any tensor, 0))."""
        steps = list(steps)
        if not steps:
            return
        cur = torch.cuda.current_stream()
        spans = []
        self._partition(steps[0][0], steps[0][1], 0)
        for n, (stream, nbytes) in enumerate(steps):
            buf = n % 2
            if n + 1 < len(steps):
                self._partition(steps[n + 1][0], steps[n + 1][1], 1 - buf)   # overlaps with the exchange below
            cur.wait_event(self.filled[buf])
            if self.used[buf]:
                cur.wait_event(self.consumed[buf])       # recv[buf] is free again
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x0.record(cur)
            got, sent_bytes = self.send[buf].exchange_into(self.recv[buf], group=self.group)
            x1.record(cur)
            spans.append((x0, x1))
            self.stats["steps"] += 1
            self.stats["link_bytes_sent"] += int(sent_bytes)
            self.stats["stream_bytes"] += int(nbytes)
            self.sent[buf].record(cur)
            self.used[buf] = True
            self.ext.wait_event(self.sent[buf])
            self.recv[buf].consume(self.graph, colour, got if self.use_v3 else self.max_tuples)
            self.consumed[buf].record(self.ext)
        self.ext.synchronize()
        cur.synchronize()
        self.stats["exchange_ms"] += sum(a.elapsed_time(b) for a, b in spans)
        for b in self.send:
            if b.overflowed():
                raise RuntimeError("an exchange bin overflowed (occurrences lost): raise its capacity")


def merge_sorted_bodies(bodies, record_size, key_bytes):
    """N-way merge of per-rank sorted .ctx bodies (disjoint key sets) into one sorted body."""
    recs = [np.frombuffer(b, dtype=np.uint8).reshape(-1, record_size) for b in bodies if len(b)]
    if not recs:
        return b""
    allr = np.concatenate(recs)
    W = key_bytes // 8
    keys = allr[:, :key_bytes].copy().view(np.uint64).reshape(-1, W)
    order = np.lexsort([keys[:, w] for w in range(W - 1, -1, -1)])
    return allr[order].tobytes()
