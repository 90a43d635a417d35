"""Host side of the .ctx v6 container (header + GraphInfo arithmetic), mirroring
src/graph/graph_writer.c:11-30,62-110 and src/basic/graph_info.c:116-175.
The C host program (mccortex_amd/host/) carries the same logic for the CLI; this
module lets tests and bench.py assemble a complete .ctx around Graph.export()."""
import struct

import numpy as np


def graph_info_update(mean, total, added_seq, num_contigs):
    """graph_info_update_contigs (graph_info.c:116-133) -> (mean, total)."""
    if not added_seq and not num_contigs:
        return mean, total
    have = 0
    if total and mean:
        have = int(float(total) / mean + 0.5)
    if have + num_contigs > 0:
        mean = int(float(total + added_seq) / (have + num_contigs)) & 0xFFFFFFFF
    return mean, total + added_seq


class CtxHeader:
    """Per-colour GraphInfo of a graph being built (graph_info.h:20-27)."""

    def __init__(self, kmer_size, ncols):
        self.k, self.ncols = kmer_size, ncols
        self.W = (2 * kmer_size + 63) // 64
        self.mean = [0] * ncols
        self.total = [0] * ncols
        self.names = ["undefined"] * ncols

    def update_stats(self, colour, total_bases_loaded, contigs_parsed):
        """graph_info_update_stats (graph_info.c:172-175): once per input file, in order."""
        self.mean[colour], self.total[colour] = graph_info_update(
            self.mean[colour], self.total[colour], total_bases_loaded, contigs_parsed)


def _x87_seq_err(total):
    """seq_err after graph_info_merge into a fresh header (graph_info.c:135-170):
    (0.01L*0 + 0.01L*T)/T in x87 long double, as 10 value bytes + 6 zero bytes.
    numpy.longdouble is the x87 80-bit type on x86-64 Linux."""
    ld = np.longdouble
    e = ld(0.01)  # (long double)(double)0.01
    if total > 0:
        e = (ld(0.01) * ld(0) + ld(0.01) * ld(total)) / ld(total)
    raw = np.array([e], dtype=np.longdouble).tobytes()
    assert len(raw) == 16, "x87 long double expected"
    return raw[:10] + b"\0" * 6


def ctx_header_bytes(h):
    out = [b"CORTEX", struct.pack("<IIII", 6, h.k, h.W, h.ncols)]
    means, totals = [], []
    for c in range(h.ncols):
        mean, total = 0, 0
        stotal, smean = h.total[c], h.mean[c]
        if stotal > 0:
            contigs = int(float(stotal) / smean + 0.5) if (stotal and smean) else 0
            mean, total = graph_info_update(0, 0, stotal, contigs)
        means.append(mean)
        totals.append(stotal)
    out.append(struct.pack("<%dI" % h.ncols, *means))
    out.append(struct.pack("<%dQ" % h.ncols, *totals))
    for c in range(h.ncols):
        nm = h.names[c].encode()
        out.append(struct.pack("<I", len(nm)) + nm)
    for c in range(h.ncols):
        out.append(_x87_seq_err(h.total[c]))
    for c in range(h.ncols):
        out.append(b"\0" * 12 + struct.pack("<I", 9) + b"undefined")
    out.append(b"CORTEX")
    return b"".join(out)


def write_ctx(path, header, body):
    with open(path, "wb") as f:
        f.write(ctx_header_bytes(header))
        f.write(body)
