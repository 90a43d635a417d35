"""Experiment: the bench's host_fed leg (10 x 5 M reads from pinned host memory through
mcx_graph_add_reads, flush inside the clock) under staging variants; one process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
if os.environ.get("PIN_NODE"):
    def cpus(n):
        out = set()
        for part in open("/sys/devices/system/node/node%%s/cpulist" %% n).read().strip().split(","):
            a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
        return out
    os.sched_setaffinity(0, cpus(os.environ["PIN_NODE"]))
import numpy as np, torch
import bench, mccortex_amd as mcx
dev = torch.device("cuda", 0)
B = 5_000_000; N = 10
genome = bench.make_genome(200_000_000, dev, 42)
hb = []
for i in range(N):
    b = bench.make_batch(genome, B, 1000 + i, dev)
    t = torch.empty((B, 150), dtype=torch.uint8).pin_memory()
    t.copy_(b.reshape(B, 151)[:, :150]); hb.append(t.numpy().reshape(-1)); del b
del genome; torch.cuda.empty_cache()
offs = np.arange(B + 1, dtype=np.uint64) * 150
g = mcx.Graph(31, 1, 1 << 30)
g.add_reads(0, hb[0][:150000], offs[:1001]); g.sync(); g.reset(); g.sync()
res = []
import resource
def thr():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return 0, 0
for rep in range(int(os.environ.get("REPS", "3"))):
    time.sleep(0.3)
    r0 = resource.getrusage(resource.RUSAGE_SELF); th0 = thr()
    t0 = time.perf_counter()
    for h in hb: g.add_reads(0, h, offs)
    t1 = time.perf_counter(); g.sync(); dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF); th1 = thr()
    res.append("%%.1f G/s (submit %%.0f of %%.0f ms; cpu %%.2f s; throttled %%d x, %%.0f ms)" %% (N * B * 120 / dt / 1e9, (t1 - t0) * 1e3, dt * 1e3,
               r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime, th1[0] - th0[0], (th1[1] - th0[1]) / 1e3))
    g.reset(); g.sync()
print("; ".join(res))
''' % ROOT
variants = [dict(), dict(MCX_IDLE_FLUSH="0"), dict(MCX_STAGE_THREADS="24")]
if len(sys.argv) > 1:  # variants from the command line: "A=1,B=2" per argument ("-" = the defaults)
    variants = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else dict() for a in sys.argv[1:]]
for env in variants:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = [l for l in p.stderr.decode().splitlines() if l.startswith("[stage]")]
    print(env, p.stdout.decode().strip().splitlines()[-1:], err[-1:], flush=True)
