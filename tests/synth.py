"""Deterministic synthetic read sets: random genome, reads at random
positions, 50% reverse-complemented, substitution errors, a few reads with N."""
import numpy as np

_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def genome(n, seed=42):
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]


def reads(nreads, readlen, genome_len=None, seed=42, err=0.001, n_frac=0.01, lower_frac=0.0,
          var_len=False, g=None):
    """-> (bases uint8[sum len], offsets uint64[nreads+1])"""
    rng = np.random.default_rng(seed + 1)
    if g is None:
        g = genome(genome_len or max(4 * readlen, nreads * readlen // 10), seed)
    G = len(g)
    lens = np.full(nreads, readlen, dtype=np.int64)
    if var_len:
        lens = rng.integers(0, 2 * readlen, nreads)
    lens = np.minimum(lens, G)
    starts = rng.integers(0, G - lens + 1)
    offs = np.zeros(nreads + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    idx = np.repeat(starts - offs[:-1].astype(np.int64), lens) + np.arange(total)
    b = g[idx].copy()
    # substitutions
    e = rng.random(total) < err
    b[e] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(e.sum()))]
    # reverse-complement half of the reads
    rc = rng.random(nreads) < 0.5
    rid = np.repeat(np.arange(nreads), lens)
    pos = np.arange(total) - offs[:-1].astype(np.int64)[rid]
    src = np.where(rc[rid], offs[:-1].astype(np.int64)[rid] + lens[rid] - 1 - pos, np.arange(total))
    b2 = b[src]
    b2 = np.where(rc[rid], _COMP[b2], b2)
    # N in a fraction of reads
    hasn = np.nonzero((rng.random(nreads) < n_frac) & (lens > 0))[0]
    if len(hasn):
        npos = offs[:-1].astype(np.int64)[hasn] + rng.integers(0, lens[hasn])
        b2[npos] = ord("N")
    if lower_frac > 0:
        low = rng.random(nreads) < lower_frac
        m = low[rid]
        b2[m] = b2[m] | 0x20
    return np.ascontiguousarray(b2, dtype=np.uint8), offs


def to_stream(bases, offs):
    """bases/offsets -> separator-delimited stream (one '\\n' after every read)."""
    n = len(offs) - 1
    lens = np.diff(offs.astype(np.int64))
    out = np.full(len(bases) + n, ord("\n"), dtype=np.uint8)
    rid = np.repeat(np.arange(n), lens)
    out[np.arange(len(bases)) + rid] = bases
    return out
