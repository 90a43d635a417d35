"""End-to-end CLI timing: write a synthetic FASTQ/FASTA, run mccortex31 build --sort, report stages."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

NR = int(os.environ.get("NR", "5000000"))
dev = torch.device("cuda", 0)
genome = bench.make_genome(100_000_000, dev, 42)
s = bench.make_batch(genome, NR, 7, dev).reshape(NR, 151).cpu().numpy()
del genome
torch.cuda.empty_cache()
out = "/tmp/e2e"
os.makedirs(out, exist_ok=True)
# FASTQ: @r\nSEQ\n+\nQUAL\n with fixed-width header "@r%08d"
hdr = np.frombuffer(b"@r00000000\n", np.uint8)
rec = np.empty((NR, 11 + 151 + 2 + 151), np.uint8)
rec[:, :11] = hdr
ids = np.arange(NR)
for d in range(8):
    rec[:, 9 - d] = 48 + (ids // 10 ** d) % 10
rec[:, 11:162] = s
rec[:, 162] = ord("+"); rec[:, 163] = ord("\n")
rec[:, 164:314] = ord("I"); rec[:, 314] = ord("\n")
fq = os.path.join(out, "reads.fq")
rec.tofile(fq)
print("wrote %s: %.2f GB, %d reads" % (fq, os.path.getsize(fq) / 1e9, NR), flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mccortex_amd", "bin", "mccortex31")
if os.environ.get("GZ"):   # GZ=1: the same reads as four .gz files: do the reader threads inflate them side by side?
    parts = []
    for i in range(4):
        pth = os.path.join(out, "part%d.fq" % i)
        rec[i * NR // 4:(i + 1) * NR // 4].tofile(pth)
        subprocess.run(["gzip", "-1", "-f", pth], check=True)
        parts.append(pth + ".gz")
    print("wrote 4 x %.2f GB .gz" % (os.path.getsize(parts[0]) / 1e9), flush=True)
    for t in ("1", "2", "4", "8"):
        t0 = time.perf_counter()
        a = [exe, "build", "-f", "-k", "31", "-n", "512M", "-m", "12G", "-s", "smp", "--sort", "-t", t]
        for pth in parts:
            a += ["--seq", pth]
        p = subprocess.run(a + [os.path.join(out, "o.ctx")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        print("build of 4 .gz files, -t %s: rc=%d wall %.2f s (%.2f G bases/s)" % (t, p.returncode, dt, NR * 150 / dt / 1e9), flush=True)
    sys.exit(0)
VARIANTS = (["--sort", "-t", "1"], ["--sort", "-t", "2"], ["--sort", "-t", "8"], ["--sort", "-t", "32"], ["-t", "32"])
if os.environ.get("PREFS"):   # e.g. PREFS=1: the read preferences that take the reads through the cutting kernels
    VARIANTS = (["--sort", "-t", "8"], ["--sort", "-t", "8", "-Q", "10"], ["--sort", "-t", "8", "-H", "8"],
                ["--sort", "-t", "8", "--remove-pcr"], ["--sort", "-t", "8", "-Q", "10", "--remove-pcr"])
for args in VARIANTS:
    t0 = time.perf_counter()
    p = subprocess.run([exe, "build", "-f", "-k", "31", "-n", "512M", "-m", "12G", "-s", "smp"] + args + ["--seq", fq, os.environ.get("OUT", os.path.join(out, "o.ctx"))],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    err = p.stderr.decode()
    lines = [l for l in err.splitlines() if "kmers" in l.lower() or "time" in l.lower() or "Dumped" in l]
    print("build %s: rc=%d wall %.2f s (%.2f G bases/s, %.2f G k-mers/s end to end)" % (" ".join(args), p.returncode, dt, NR * 150 / dt / 1e9, NR * 120 / dt / 1e9))
    for l in lines[-4:]:
        print("   ", l[-150:])
print("ctx size %.2f GB" % (os.path.getsize(os.path.join(out, "o.ctx")) / 1e9))
# stages of one run, from the status lines
p = subprocess.run([exe, "build", "-f", "-k", "31", "-n", "512M", "-m", "12G", "-s", "smp", "--sort", "-t", "32", "--seq", fq, os.path.join(out, "o.ctx")],
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
for l in p.stderr.decode().splitlines():
    print("   ", l[:160])
