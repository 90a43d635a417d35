"""The host entry at a size where its background flush runs: reads in host memory ->
mcx_graph_add_reads (staging threads, packed chunks, H2D) with the idle-device flush forced at every
chunk (MCX_IDLE_FLUSH=2: one region group per chunk, csrc/mcx_api.hip flush_if_device_idle) must
build the graph the device-resident path builds.  Replaces the worker loop of
src/basic/async_read_io.c:283-310 + src/tools/build_graph.c:233-254."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import bench, mccortex_amd as mcx
dev = torch.device("cuda", 0)
B, L = 2_000_000, 150
genome = bench.make_genome(20_000_000, dev, 7)
b = bench.make_batch(genome, B, 77, dev)
host = b.reshape(B, L + 1)[:, :L].contiguous().cpu().numpy().reshape(-1)
offs = np.arange(B + 1, dtype=np.uint64) * L
res = []
for mode in ("host", "dev"):
    g = mcx.Graph(31, 1, 1 << 27)
    if DEFER:
        g.configure("defer_tuples", DEFER)
    g.configure("profile", 1)
    for rep in range(3):
        if mode == "host":
            g.add_reads(0, host, offs)
        else:
            g.add_stream_dev(0, b, b.numel())
    g.sync()
    st = g.device_stats()
    cs, n = g.checksum()
    prof = g.profile()
    res.append((cs, n, st.num_kmers_loaded, st.contigs_parsed, prof["k_lds_insert"][0]))
    g.close()
print("RESULT", res)
'''


def _run(env, defer=0):
    p = subprocess.run([sys.executable, "-c", "DEFER = %d\n" % defer + CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("RESULT")][-1]
    return eval(line[len("RESULT"):])


def test_host_fed_build_with_background_flush_matches_device_resident_build():
    env = dict(os.environ, MCX_IDLE_FLUSH="2", MCX_STAGE_BYTES=str(32 << 20))  # 20 chunks per call: 20 chances to flush a group
    host, dev = _run(env)
    assert host[:4] == dev[:4]          # checksum, nodes, k-mers loaded, contigs
    assert host[2] > 700_000_000 and host[1] > 10_000_000
    assert host[4] > dev[4]             # ... and the background flush did run (more LDS-insert launches than one closing flush)


def test_idle_flushes_that_start_on_a_nearly_full_workspace():
    """The device was busy first: the idle flushes only start once 3/4 of the flush size is buffered
    (MCX_IDLE_FLUSH=3), with a flush size the three calls exceed.  A flushed region group must not make room
    in the books for the groups that have not been emptied yet (flush_if_device_idle: `pending` follows the
    group that has waited longest): same graph as the device-resident build, whole flushes in between included."""
    env = dict(os.environ, MCX_IDLE_FLUSH="3", MCX_STAGE_BYTES=str(32 << 20))
    host, dev = _run(env, defer=400_000_000)   # 3 x 240 M occurrences through a 400 M workspace
    assert host[:4] == dev[:4]
    assert host[2] > 700_000_000


def test_host_entry_without_the_one_pass_packer():
    """Hosts without AVX-512 VBMI2 assemble 16 KiB blocks of the stream as ASCII and pack those (MCX_FUSED_PACK=0 forces
    that path; the SWAR packer with MCX_NO_AVX2=1): same graph as the device-resident build."""
    for extra in ({"MCX_FUSED_PACK": "0"}, {"MCX_NO_AVX2": "1"}):
        env = dict(os.environ, MCX_STAGE_BYTES=str(32 << 20), **extra)
        host, dev = _run(env)
        assert host[:4] == dev[:4], extra
