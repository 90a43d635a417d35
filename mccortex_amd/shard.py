"""Hash-prefix sharded build over torch.distributed (RCCL on the GPU box, gloo in CPU tests).

The data path is exchange-only: every rank bins its per-occurrence tuples by owner and one
all-to-all moves each bin to the rank that owns it.  `exchange()` is backend-agnostic host logic;
the device kernels are reached through the Graph handle (mccortex_amd/graph.py)."""
import numpy as np
import torch
import torch.distributed as dist


def exchange(send_keys, send_edges, counts, group=None):
    """send_keys [world, cap, W] int64, send_edges [world, cap] uint8, counts [world] int64 (same
    device).  Returns (recv_keys [n, W], recv_edges [n], recv_counts list) for this rank."""
    world = dist.get_world_size(group)
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts, group=group)
    sc, rc = counts.tolist(), recv_counts.tolist()
    ro = np.concatenate([[0], np.cumsum(rc)]).astype(np.int64)
    W = send_keys.shape[2]
    n_in = int(ro[-1])
    recv_keys = torch.empty((n_in, W), dtype=send_keys.dtype, device=send_keys.device)
    recv_edges = torch.empty((n_in,), dtype=send_edges.dtype, device=send_edges.device)
    # bins are fixed-capacity: pack the filled prefixes, then one all_to_all_single per array
    # (works on RCCL and gloo alike; every pair of GPUs is one xGMI hop)
    pk = torch.cat([send_keys[p, :sc[p]] for p in range(world)])
    pe = torch.cat([send_edges[p, :sc[p]] for p in range(world)])
    dist.all_to_all_single(recv_keys, pk, output_split_sizes=rc, input_split_sizes=sc, group=group)
    dist.all_to_all_single(recv_edges, pe, output_split_sizes=rc, input_split_sizes=sc, group=group)
    return recv_keys, recv_edges, rc


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
