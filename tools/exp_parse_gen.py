"""synthetic FASTQ for the host-side experiments: N reads x 150 bp drawn from a 200 Mbp random genome (argv[2] = 'iid':
every read random, i.e. every k-mer novel), constant quality, 4-line records"""
import numpy as np, sys
n = int(sys.argv[1]); iid = len(sys.argv) > 2 and sys.argv[2] == "iid"
rng = np.random.default_rng(1)
L = 150
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
genome = None if iid else acgt[rng.integers(0, 4, 200_000_000, dtype=np.uint8)]
with open("reads.fq", "wb") as f:
    for lo in range(0, n, 2_000_000):
        m = min(2_000_000, n - lo)
        rec = np.empty((m, 3 + L + 1 + 2 + L + 1), dtype=np.uint8)
        rec[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        if iid:
            rec[:, 3:3 + L] = acgt[rng.integers(0, 4, (m, L), dtype=np.uint8)]
        else:
            pos = rng.integers(0, len(genome) - L, m)
            rec[:, 3:3 + L] = genome[pos[:, None] + np.arange(L)[None, :]]
        rec[:, 3 + L] = 10
        rec[:, 4 + L:6 + L] = np.frombuffer(b"+\n", dtype=np.uint8)
        rec[:, 6 + L:-1] = ord("I"); rec[:, -1] = 10
        rec.tofile(f)
