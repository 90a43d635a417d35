"""End-to-end CLI timing on the bench's 10 M-read FASTQ (C2 shape, 200 Mbp genome): the python parent
frees the GPU before the runs.  Variants through the environment: MCX_PAR_BATCH, -t."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
B = 5_000_000
genome = bench.make_genome(200_000_000, dev, 42)
out = "/tmp/e2e2"; os.makedirs(out, exist_ok=True)
fq = os.path.join(out, "reads.fq")
hdr = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device=dev); mid = torch.tensor(list(b"+\n"), dtype=torch.uint8, device=dev)
with open(fq, "wb") as f:
    for i in range(2):
        b = bench.make_batch(genome, B, 1000 + i, dev)
        rec = torch.empty((B, 3 + 151 + 2 + 151), dtype=torch.uint8, device=dev)
        rec[:, :3] = hdr; rec[:, 3:154] = b.reshape(B, 151); rec[:, 154:156] = mid; rec[:, 156:-1] = ord("I"); rec[:, -1] = ord("\n")
        rec.cpu().numpy().tofile(f); del rec, b
del genome; torch.cuda.empty_cache()
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mccortex_amd", "bin", "mccortex31")
for env, t in [({}, "32"), ({}, "32"), ({}, "32"), ({}, "16"), ({}, "48"), ({"MCX_STAGE_THREADS": "8"}, "32")]:
    t0 = time.perf_counter()
    p = subprocess.run([exe, "build", "-f", "-k", "31", "-n", "1G", "-m", "24G", "-t", t, "--sort", "-s", "x", "--seq", fq, os.path.join(out, "o.ctx")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MCX_TIMING="1", **env))
    dt = time.perf_counter() - t0
    st = [l.split("ms", 1) for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and "epoch" not in l and "export" not in l]
    print("%-34s -t %-3s rc=%d wall %.3f s = %.2f G k-mers/s | %s" % (env, t, p.returncode, dt, 1.2 / dt,
          "; ".join("%s %s" % (b.strip()[:22], a.replace("[timing]", "").strip()) for a, b in st)), flush=True)
