"""Sequence files -> the device byte stream of mcx_graph_add_stream_dev (reads separated by at least
one byte that is not ACGTacgt), for the multi-GPU build tool (mgpu_build.py).

Plumbing only (torch ops on whatever device the bytes live on): every rank takes a byte range of
the file, moves its ends to record boundaries, uploads the raw bytes and drops everything that is
not sequence.  Formats the reference's `build` is fed with: FASTA (multi-line
sequences allowed), FASTQ (4-line records), "plain" (one sequence per line); uncompressed regular
files, records shorter than a step."""
import os

import numpy as np
import torch

NL, CR, GT, AT, PLUS = 10, 13, ord(">"), ord("@"), ord("+")


def detect_format(path):
    """first non-blank byte: '>' FASTA, '@' FASTQ, else plain (seq_file's own sniffing rule)"""
    with open(path, "rb") as f:
        head = f.read(4096)
    if head[:2] == b"\x1f\x8b":
        raise ValueError("%s: gzip input is not supported by the multi-GPU tool (byte ranges)" % path)
    for c in head:
        if c in (NL, CR, 32, 9):
            continue
        return "fasta" if c == GT else "fastq" if c == AT else "plain"
    return "plain"


def _line_starts(win, base):
    """absolute offsets of the line starts inside the window (a line starts after every newline)"""
    nl = np.flatnonzero(win == NL)
    return nl + 1 + base


def next_record_start(f, size, pos, fmt, window=1 << 16):
    """smallest record start >= pos (size if none).  pos == 0 is a record start."""
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    while True:
        lo = pos - 1                      # the byte before pos tells whether pos starts a line
        f.seek(lo)
        win = np.frombuffer(f.read(min(window, size - lo)), dtype=np.uint8)
        starts = _line_starts(win, lo)    # line starts in (lo, lo + len(win)]
        starts = starts[starts < lo + len(win)]
        at_end = lo + len(win) >= size
        first = lambda s: win[s - lo]
        if fmt == "plain":
            if len(starts):
                return int(starts[0])
        elif fmt == "fasta":
            for s in starts:
                if first(s) == GT:
                    return int(s)
        else:  # fastq: '@' line whose second-next line starts with '+' and whose fourth-next with '@' (or ends the file)
            for i, s in enumerate(starts):
                if first(s) != AT:
                    continue
                if i + 2 < len(starts) and first(starts[i + 2]) == PLUS and (
                        (i + 4 < len(starts) and first(starts[i + 4]) == AT) or (i + 4 >= len(starts) and at_end)):
                    return int(s)
                if i + 4 >= len(starts) and not at_end:
                    break                 # cannot tell inside this window: grow it
        if at_end:
            return size
        window *= 4


def plan_steps(path, fmt, rank, world, step_bytes):
    """[(start, end)] byte ranges of this rank's share of the file, each starting at a record start
    and at most ~step_bytes long (a record is never split: longer records are an error)"""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        a = next_record_start(f, size, size * rank // world, fmt)
        b = next_record_start(f, size, size * (rank + 1) // world, fmt)
        steps, s = [], a
        while s < b:
            e = min(b, s + step_bytes)
            if e < b:
                e2 = next_record_start(f, size, e, fmt)
                e = min(e2, b)
                # the step would end beyond the buffers' size if the record at `e` is a long one: step back
                if e - s > step_bytes + (1 << 16):
                    raise ValueError("%s: a record of more than %d bytes near offset %d (use the single-GPU build)" % (path, step_bytes, s))
            steps.append((s, e))
            s = e
    return steps


def load_bytes(path, start, end, device):
    if end <= start:
        return torch.zeros(0, dtype=torch.uint8, device=device)
    a = np.fromfile(path, dtype=np.uint8, count=end - start, offset=start)
    return torch.from_numpy(a).to(device)


def to_stream(buf, fmt):
    """raw bytes of whole records (starting at a record start) -> separator stream (uint8 tensor)"""
    n = buf.numel()
    if n == 0 or fmt == "plain":
        return buf                                       # newlines are the separators already
    nl = buf == NL
    line_start = torch.ones(n, dtype=torch.bool, device=buf.device)
    line_start[1:] = nl[:-1]
    seqchar = ~nl & (buf != CR)
    if fmt == "fasta":
        # header lines go except for their '>' (the separator); sequence lines lose their newlines,
        # so a sequence spread over several lines stays ONE read
        hdr_start = line_start & (buf == GT)
        line_id = torch.cumsum(line_start, 0, dtype=torch.int32) - 1
        nlines = int(line_id[-1].item()) + 1
        is_hdr = torch.zeros(nlines, dtype=torch.bool, device=buf.device)
        is_hdr[line_id[hdr_start].long()] = True
        in_hdr = is_hdr[line_id.long()]
        keep = (seqchar & ~in_hdr) | hdr_start
        return buf[keep]
    # fastq, 4-line records: line 0 '@name' (its '@' stays as the separator), line 1 bases, line 2 '+', line 3 qualities
    line_id = torch.cumsum(line_start, 0, dtype=torch.int32) - 1
    phase = line_id & 3
    first = line_start & (buf != NL)
    bad = (first & (phase == 0) & (buf != AT)) | (first & (phase == 2) & (buf != PLUS))
    if bool(bad.any().item()):
        at = int(torch.nonzero(bad)[0].item())
        raise ValueError("irregular FASTQ record near byte %d of the step (multi-line FASTQ is not supported here)" % at)
    keep = (seqchar & (phase == 1)) | (line_start & (phase == 0))
    return buf[keep]
