"""World-size-2 gloo test of the sharded build's host logic (exchange + merge), on CPU.
Tuples come from the oracle (allowed in tests); owners from the product's mcx_key_owner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mccortex_amd as mcx
    from mccortex_amd import shard
    from oracle import orc
    shard.MAX_ROUND = 30000   # several rounds; per-rank counts differ, the round count must not
    g = synth.genome(20000, 1)
    bases, offs = synth.reads(1500, 100, seed=100 + rank, g=g, n_frac=0.05)
    keys, edges = orc.tuples(k, bases, offs)
    W = keys.shape[1]
    owner = np.array([mcx.key_owner([int(x) for x in row], k, world) for row in keys], dtype=np.int64)
    cap = len(keys)
    sk = torch.zeros((world, cap, W), dtype=torch.int64)
    se = torch.zeros((world, cap), dtype=torch.uint8)
    cnt = torch.zeros(world, dtype=torch.int64)
    for p in range(world):
        m = owner == p
        n = int(m.sum())
        sk[p, :n] = torch.from_numpy(keys[m].view(np.int64))
        se[p, :n] = torch.from_numpy(edges[m])
        cnt[p] = n
    rk, re, rc = shard.exchange(sk, se, cnt)
    rkeys = rk.numpy().view(np.uint64)
    # everything received is owned by this rank
    assert all(mcx.key_owner([int(x) for x in row], k, world) == rank for row in rkeys[::97])
    K, C, E = orc.graph_from_tuples(rkeys, re.numpy())
    rs = 8 * W + 5
    body = np.zeros((len(K), rs), dtype=np.uint8)
    body[:, :8 * W] = K.view(np.uint8).reshape(len(K), 8 * W)
    body[:, 8 * W:8 * W + 4] = C[:, :1].copy().view(np.uint8).reshape(len(K), 4)
    body[:, 8 * W + 4] = E[:, 0]
    np.save(os.path.join(tmpdir, "body%d.npy" % rank), body)
    np.save(os.path.join(tmpdir, "in%d.npy" % rank), np.concatenate([bases, np.zeros(1, np.uint8)]))
    np.save(os.path.join(tmpdir, "off%d.npy" % rank), offs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [31, 47])
def test_two_rank_exchange_equals_single_graph(mcx, orc, tmp_path, k):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), k, str(tmp_path)), nprocs=world, join=True)
    from mccortex_amd import shard
    W = (2 * k + 63) // 64
    rs = 8 * W + 5
    bodies = [np.load(tmp_path / ("body%d.npy" % r)).tobytes() for r in range(world)]
    merged = shard.merge_sorted_bodies(bodies, rs, 8 * W)
    og = orc.Graph(k, 1, 1 << 20)
    for r in range(world):
        b = np.load(tmp_path / ("in%d.npy" % r))[:-1]
        o = np.load(tmp_path / ("off%d.npy" % r))
        og.add_reads(0, b, o)
    assert merged == og.ctx_bytes(True)[og.header_size():]


def _block_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mccortex_amd import shard
    shard.MAX_BLOCK_BYTES = 4096  # several rounds over the segment axis
    segs, seg_cap, ov_cap, W = 24, 50, 33, 2

    def stamp(src, dst):  # what rank `src` addresses to rank `dst`
        g = torch.Generator().manual_seed(1000 * src + dst)
        return (torch.randint(-2**62, 2**62, (segs, seg_cap, W), generator=g),
                torch.randint(0, seg_cap + 5, (segs,), generator=g),
                torch.randint(-2**62, 2**62, (ov_cap, W), generator=g),
                torch.randint(0, 256, (ov_cap,), generator=g).to(torch.uint8),
                torch.randint(0, ov_cap, (1,), generator=g))

    send = shard.BlockExchange(world, segs, seg_cap, ov_cap, W, "cpu")
    recv = shard.BlockExchange(world, segs, seg_cap, ov_cap, W, "cpu")
    for p in range(world):
        k, c, ok, oe, oc = stamp(rank, p)
        send.keys[p], send.counts[p], send.ov_keys[p], send.ov_edges[p], send.ov_counts[p] = k, c, ok, oe, oc[0]
    send.exchange_into(recv)
    for p in range(world):
        k, c, ok, oe, oc = stamp(p, rank)
        assert torch.equal(recv.keys[p], k) and torch.equal(recv.counts[p], c)
        assert torch.equal(recv.ov_keys[p], ok) and torch.equal(recv.ov_edges[p], oe)
        assert int(recv.ov_counts[p]) == int(oc[0])
    assert not send.overflowed()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_block_exchange_routes_fixed_blocks(mcx, world):
    mp.spawn(_block_worker, args=(world, _free_port()), nprocs=world, join=True)


def _superk_worker(rank, world, port, rec_words=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mccortex_amd import shard
    segs, seg_cap = 8, 300

    def stamp(src, dst):  # what rank `src` addresses to rank `dst`: fills and records
        g = torch.Generator().manual_seed(7000 * src + dst)
        cnt = torch.randint(0, seg_cap + 1, (segs,), generator=g)
        cnt[0] = 0
        cnt[1] = seg_cap
        return cnt, torch.randint(-2**62, 2**62, (segs, seg_cap, rec_words), generator=g)

    send = shard.SuperkExchange(world, segs, seg_cap, "cpu", rec_words=rec_words)
    recv = shard.SuperkExchange(world, segs, seg_cap, "cpu", rec_words=rec_words)
    for p in range(world):
        send.fills[:, p], send.recs[p] = stamp(rank, p)
    n, sent = send.exchange_into(recv)
    assert sent == sum(int(stamp(rank, p)[0].sum()) for p in range(world) if p != rank) * rec_words * 8 + (world - 1) * segs * 8
    tot = 0
    for p in range(world):
        cnt, recs = stamp(p, rank)
        assert torch.equal(recv.counts[p], cnt)
        for s_ in range(segs):
            assert torch.equal(recv.recs[p, s_, :cnt[s_]], recs[s_, :cnt[s_]])
        tot += int(cnt.sum())
    assert n == tot and not send.overflowed()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rec_words", [(2, 2), (4, 2), (2, 4)])
def test_superk_exchange_routes_filled_parts(mcx, world, rec_words):
    """rec_words 2: 16-byte records of one-word keys; 4: the 32-byte records of two-word keys"""
    mp.spawn(_superk_worker, args=(world, _free_port(), rec_words), nprocs=world, join=True)
