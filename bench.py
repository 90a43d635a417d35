#!/usr/bin/env python3
"""bench.py -- k-mers/s inserted by the `build` hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): k=31, 1 colour, synthetic 150 bp reads drawn from a
200 Mbp random genome (50% reverse strand, 0.1% substitutions, 1% of reads with an N), table of
2^30 slots.  One step = one batch of 5,000,000 reads (600M k-mer occurrences) that is already
resident in HBM as a '\\n'-separated ASCII byte stream (`value`: the 2-bit packing of the bases is
INSIDE the clock -- the kernel encodes the ASCII bytes in its tile prologue; parse and H2D are outside:
see `host_fed` and `e2e`); 10 steps are the 50M x 150bp set.  The same run also reports, as separate objects of the
JSON line: `host_fed` (the same reads handed over in host memory: staging + PCIe inside the clock),
`e2e` (`mccortex31 build --sort` on a FASTQ file of the same shape: process start, parse, build, sort,
.ctx write), `default_defer` (the library's own flush size instead of the bench's), `other_configs`
(C4: k=63; C5-like: 4 colours; C2-stress: iid reads; `hashtest`: the reference's own published table benchmark),
`e2e_full` (the CLI on the whole 50M x 150bp FASTQ) and `cpu_baseline` (the oracle on the host cores, on the SAME
reads and the same `-n 1G` table; its records' checksum must equal the GPU graph's: `config.checksum_matches_oracle`).
`roofline` carries the device's MEASURED ceilings (streaming copy, random 64-byte-sector RMW over a table-sized
working set: mcx_ubench_*) next to the nominal 8 TB/s, and per-kernel durations from a second, non-overlapped pass.

N>1 (one process per GPU, torchrun).  Default `--scaling strong` = BASELINE config C3: the SAME reads
as N=1 (step i is the same 5M-read batch, every rank takes reads [r B/N, (r+1) B/N) of it), the same
200 Mbp genome, a table of 2^30 slots IN TOTAL (2^30 / N per GPU); the sum of the ranks' graph
checksums must equal the N=1 `graph_checksum` (`config.checksum_matches_n1`).  `--scaling weak`: every
rank takes its own 5M-read batch per step (genome scaled to N x 200 Mbp, 2^30 slots per GPU).  Either
way a rank cuts its reads into per-owner super-k-mer records (owner = hash of the k-mer's canonical
minimizer), exchanges them with RCCL all-to-alls and k-merises + inserts what it owns (exchange
format v3).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 31
READ_LEN = int(os.environ.get("MCX_BENCH_READ_LEN", "150"))  # (experiments only: C2 is 150)
BATCH_READS = 5_000_000
GENOME_PER_GPU = 200_000_000
TABLE_SLOTS = 1 << 30
ALG_BYTES_PER_KMER = 21.25   # 1.25 input + 8 key + 8 covg RMW + ~4 edge RMW
ALG_BYTES_PER_NOVEL = 8.0    # key write when the node is new
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
DEFER_TUPLES = 8_000_000_000  # k-mer occurrences buffered per flush of the partition->LDS-insert path (the least `value` uses)
PLACE_BINS = 16  # mcx_graph_configure("place_bins"): each half of the sub-table bins is the best of up to this many allocations by the
                # split's write-pattern probe (where 8 GB of bins lie in HBM decides 13 % of the split: profiles/r06_experiments.md)
HEADLINE_DEFER = DEFER_TUPLES  # what `value` was measured with (main() sets it: the whole timed region in one flush)
# algorithmic bytes per k-mer occurrence of every kernel of that path
KERNEL_ALG_BYTES = {"k_stream": 21.25, "k_stream_bin": 1.25 + 8.0, "k_tuples_bin": 16.0, "k_lds_insert": 8.0,
                    "k_insert_tuples": 29.0, "k_stream_superk": 1.25 + 2.3, "k_superk_bin": 2.3 + 8.0}


# graph_checksum of the default workload (B = 5M reads per step, 200 Mbp genome, err 0.001) after
# `steps` steps, as N=1 runs of this file report it (BENCH_r02.json: 20 steps; profiles/r02g: 10)
N1_CHECKSUMS = {10: "c72ff066d6a527ec", 20: "bc686b82dd898aa5"}
# the same for the C5 pass (4 colours, reads [c T/4, (c+1) T/4) of the timed reads to sample c): config.C5_reference of N=1 runs
N1_C5_CHECKSUMS = {}


def make_genome(n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    return lut[torch.randint(0, 4, (n,), generator=g, device=device)]


def make_batch(genome, nreads, seed, device, err_rate=0.001):
    """-> uint8 [nreads*(READ_LEN+1)] stream, every read followed by '\\n'."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((nreads, READ_LEN + 1), dtype=torch.uint8, device=device)
    ar = torch.arange(READ_LEN, device=device)
    sub = 500_000
    for lo in range(0, nreads, sub):
        n = min(sub, nreads - lo)
        starts = torch.randint(0, genome.numel() - READ_LEN, (n, 1), generator=g, device=device)
        r = genome[starts + ar]
        err = torch.rand((n, READ_LEN), generator=g, device=device) < err_rate
        r = torch.where(err, acgt[torch.randint(0, 4, (n, READ_LEN), generator=g, device=device)], r)
        rc = torch.rand((n, 1), generator=g, device=device) < 0.5
        r = torch.where(rc, comp[r.flip(1).long()], r)
        hasn = torch.rand((n,), generator=g, device=device) < 0.01
        npos = torch.randint(0, READ_LEN, (n,), generator=g, device=device)
        rows = torch.nonzero(hasn).squeeze(1)
        r[rows, npos[rows]] = ord("N")
        out[lo:lo + n, :READ_LEN] = r
    out[:, READ_LEN] = ord("\n")
    return out.reshape(-1)


def make_batch_iid(nreads, seed, device):
    """C2-stress: iid random reads, (nearly) every k-mer novel."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((nreads, READ_LEN + 1), dtype=torch.uint8, device=device)
    sub = 500_000
    for lo in range(0, nreads, sub):
        n = min(sub, nreads - lo)
        out[lo:lo + n, :READ_LEN] = acgt[torch.randint(0, 4, (n, READ_LEN), generator=g, device=device)]
    out[:, READ_LEN] = ord("\n")
    return out.reshape(-1)


def csrc_digest():
    """sha1 over the sources of the three build kernels (k_stream_bin, k_tuples_bin, k_lds_insert and what they
    include): a PMC traffic figure is only valid for the kernels it was taken on.  (Until round 4 the digest covered
    every file under csrc/, so host-side edits -- staging threads, the multi-GPU facade -- voided it as well.)"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mccortex_amd", "csrc")
    for n in ("mcx_defer.h", "mcx_kernels.h", "mcx_kmer.h"):
        h.update(n.encode())
        h.update(open(os.path.join(d, n), "rb").read())
    # ... and their launch geometry (grids, block sizes, LDS sizes, variant choice): the launch functions of mcx_api.hip
    # (round 4's last change -- 8 blocks per CU for the 512-thread k-merising kernel -- lived there and escaped the digest)
    api = open(os.path.join(d, "mcx_api.hip"), "rb").read()
    lo, hi = api.find(b"static void launch_bin_stream_pk"), api.find(b"static void free_defer(mcx_graph *g)\n{")
    if lo < 0 or hi < lo:
        raise RuntimeError("csrc_digest: launch functions not found in mcx_api.hip")
    h.update(b"launch-geometry")
    h.update(api[lo:hi])
    return h.hexdigest()[:16]


def pmc_traffic(kernel, occurrences_per_launch):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json,
    written by tools/prof.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, corrected as MI355X_MICROARCH.md prescribes), scaled to this run's launch size.  Refused
    (None) when the summary was taken on other kernel sources than the ones that just ran."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        meta = json.load(open(files[-1]))
        rel = os.path.relpath(files[-1], ROOT)
        if meta.get("csrc_digest") != csrc_digest():
            return None, "%s is stale (kernel sources changed since it was taken)" % rel
        t = meta["kernels"][kernel]
        per_occ = (t["read_bytes"] + t["written_bytes"]) / t["occurrences"]
        return per_occ * occurrences_per_launch, rel
    except (KeyError, ValueError, ZeroDivisionError):
        return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_full_size(batches_dev, nsteps, table_slots, nthreads):
    """The oracle on the SAME reads and the same `-n` as the GPU build: the first `nsteps`
    steps of the workload into a table of `table_slots` slots (2^30: 14 GB of keys, coverages, edges, bucket
    bytes), timed; then the order-independent checksum of its records (mcx_records_checksum over the unsorted
    .ctx body the oracle writes) for comparison with the GPU graph of the same steps."""
    from oracle import orc
    import mccortex_amd as mcx
    g = orc.Graph(K, 1, table_slots)
    g.tune(nthreads)  # prefault with the worker threads, huge pages, per-worker node tallies (outside the clock, like calloc + first touch)
    kmers, dt = 0, 0.0
    for b in batches_dev[:nsteps]:
        n = b.numel() // (READ_LEN + 1)
        host = b.reshape(n, READ_LEN + 1)[:, :READ_LEN].contiguous().cpu().numpy().reshape(-1)
        offs = np.arange(n + 1, dtype=np.uint64) * READ_LEN
        t0 = time.perf_counter()
        st = g.add_reads(0, host, offs, nthreads=nthreads)
        dt += time.perf_counter() - t0
        kmers += st.num_kmers_loaded
        del host
    t0 = time.perf_counter()
    body = g.body_array(False)
    cs = mcx.records_checksum(body, K, 1)
    t_cs = time.perf_counter() - t0
    nodes = g.nkmers
    del body, g
    return {"kmers_per_s": kmers / dt, "seconds": dt, "kmers": int(kmers), "nodes": int(nodes), "checksum": "%016x" % cs,
            "threads": nthreads, "steps": nsteps, "table_slots": table_slots, "dump_and_checksum_s": round(t_cs, 2)}


def cpu_baseline(stream_dev, fastq_sample=None, batches_dev=None, oracle_steps=0, table_slots=TABLE_SLOTS):
    """The CPU oracle (port of the reference algorithm: bucket-locked table, pthreads) on the host's cores.
    (a) thread scan: insert phase from memory at -t {1, 8, 32, nproc} on 500k reads per point (the arrays are
    prefaulted by the same number of threads first, see orc_graph_tune: first-touch faults made the short
    samples of round 1 scale negatively); (b) `value`: the best thread count on the first `oracle_steps` steps of
    the SAME workload into the SAME `-n 1G` table as the GPU build (oracle_full_size; BASELINE.md section 3: "same
    input files and -n"), whose records' checksum bench.py compares with the GPU graph of those steps; (c) the
    reference-shaped end-to-end run -- ONE reader thread parsing the FASTQ file into a 2048-slot pool, -t
    workers (src/basic/async_read_io.c:145-175,283-310) -- on a 500k-read file."""
    from oracle import orc
    ncores = os.cpu_count() or 1
    nsample = 500_000

    s = stream_dev[:nsample * (READ_LEN + 1)].reshape(nsample, READ_LEN + 1)[:, :READ_LEN].contiguous().cpu().numpy()
    bases = s.reshape(-1)
    offs = (np.arange(nsample + 1, dtype=np.uint64) * READ_LEN)

    def run(nt):
        g = orc.Graph(K, 1, max(1 << 20, nsample * 320))  # occupancy ~0.4 as in C2
        g.tune(nt)
        t0 = time.perf_counter()
        st = g.add_reads(0, bases, offs, nthreads=nt)
        dt = time.perf_counter() - t0
        return st.num_kmers_loaded, dt

    scan, kmers = {}, 0
    for nt in sorted({t for t in (1, 8, 32, ncores) if t <= ncores}):
        kmers, dt = run(nt)
        scan[nt] = kmers / dt
    best = max(scan, key=scan.get)
    eff = effective_cores()
    out = {"value": scan[best], "unit": "k-mers/s", "cores": best, "threads": best,
           "effective_cores": round(eff, 2) if eff else ncores,
           "cores_note": "`cores` = `threads` = worker threads of the timed run (the best of the scan); `effective_cores` = the CPU time per second the container's "
                         "cgroup grants this job (cpu.max quota / period%s), which is what the threads share; host_cpus = CPUs visible"
                         % ("" if eff else "; no quota set here: all visible CPUs"),
           "kind": "port",
           "sample": "first %d reads of step 0 (%d k-mer occurrences) per thread count, oracle/mcx_oracle.c "
                     "bucket-locked table build from memory, arrays prefaulted; value = best thread count" % (nsample, kmers),
           "host_cpus": ncores, "cpu_model": cpu_model(),
           "threads_kmers_per_s": {str(k): round(v) for k, v in sorted(scan.items())}}
    if oracle_steps and batches_dev is not None:
        try:
            full = oracle_full_size(batches_dev, oracle_steps, table_slots, best)
            out["thread_scan_value"] = out["value"]
            out["value"] = full["kmers_per_s"]
            out["sample"] = ("the first %d step(s) of the SAME workload (%d reads x %d bp each, %d k-mer occurrences) into the SAME table size as the GPU "
                             "build (-n %d slots), oracle/mcx_oracle.c bucket-locked table build from memory at the best thread count of the scan "
                             "(threads_kmers_per_s: 500k reads per point into a small table), arrays prefaulted"
                             % (oracle_steps, batches_dev[0].numel() // (READ_LEN + 1), READ_LEN, full["kmers"], table_slots))
            out["full_size"] = full
        except Exception as e:  # (host memory: the table needs 14 GB)
            out["full_size"] = {"error": str(e)[:300]}
    if fastq_sample:
        e2e = {}
        for nt in sorted({t for t in (best,) if t <= ncores}):
            g = orc.Graph(K, 1, max(1 << 20, nsample * 320))
            g.tune(nt)
            tot, ins, st = g.build_file(fastq_sample, nt)
            t0 = time.perf_counter()
            g.ctx_bytes(False)  # graph_write_all_kmers_direct in table order (the reference's default: no --sort)
            tot += time.perf_counter() - t0
            e2e[str(nt)] = {"kmers_per_s": round(st.num_kmers_loaded / tot), "seconds": round(tot, 3), "insert_seconds": round(ins, 3)}
        out["e2e_reference_shaped"] = {"what": "1 reader thread per file -> 2048-slot pool -> -t workers, unsorted .ctx dump to memory; %d-read FASTQ" % nsample,
                                       "by_workers": e2e}
    return out


def pack_batches(mcx, batches):
    """device-resident ASCII streams -> their packed form (code words, invalid flags), outside any clock"""
    import torch
    out = []
    for b in batches:
        n = b.numel()
        nch = (n + 15) // 16
        code = torch.empty(nch, dtype=torch.int32, device=b.device)
        inv = torch.empty(nch, dtype=torch.int16, device=b.device)
        mcx.pack_stream_dev(b, n, code, inv)
        out.append((code, inv, n))
    torch.cuda.synchronize()
    return out


def kernel_table(prof, kmers, W=1, bases_per_kmer=1.25):
    """{kernel: launches / ms / achieved GB/s / frac of HBM peak} from the library's HIP-event spans;
    algorithmic bytes per occurrence of every kernel as in DESIGN.md section 4"""
    alg = {"k_stream": bases_per_kmer + 20.0, "k_stream_bin": bases_per_kmer + 8.0 * W, "k_tuples_bin": 16.0 * W,
           "k_lds_insert": 8.0 * W, "k_insert_tuples": 21.0 + 8.0 * W, "k_stream_superk": bases_per_kmer + 2.3, "k_superk_bin": 2.3 + 8.0}
    out = {}
    for n, (c, t) in prof.items():
        if t <= 0:
            continue
        ach = alg.get(n, 0.0) * kmers / (t * 1e-3) / 1e9
        out[n] = {"launches": c, "total_ms": round(t, 3), "avg_ms": round(t / c, 4), "achieved": round(ach, 1),
                  "frac": round(ach / HBM_PEAK_GBS, 4)}
    return out


def run_config(mcx, batches, k, ncols, colours, table_slots, defer_tuples, packed=None, cfg=None, checksum=False, isolated=False):
    """one device-resident build of `batches` (fresh graph; `packed`: their packed form) -> record.
    isolated: a second pass over the same input with the flush overlap off (one kernel at a time on one stream) gives
    per-kernel durations that mean something; `value` is the first pass (overlap on, the default)."""
    import torch
    g = mcx.Graph(k, ncols, table_slots)
    g.configure("place_bins", PLACE_BINS)
    if defer_tuples:
        g.configure("defer_tuples", defer_tuples)
    for key, v in (cfg or {}).items():
        g.configure(key, v)
    g.add_stream_dev(0, batches[0][:1024 * (READ_LEN + 1)], 1024 * (READ_LEN + 1))
    g.sync(); g.reset(); g.sync()

    def one_pass():
        g.configure("profile", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if packed is not None:
            for c, (code, inv, n) in zip(colours, packed):
                g.add_packed_dev(c, code, inv, n)
        else:
            for c, b in zip(colours, batches):
                g.add_stream_dev(c, b, b.numel())
        g.sync()
        return time.perf_counter() - t0

    dt = one_pass()
    st = g.device_stats()
    prof = g.profile()
    cs = g.checksum() if (checksum or isolated) else None
    ist = g.insert_stats()
    prof_iso, dt_iso = None, None
    if isolated:
        g.reset()
        g.configure("flush_overlap", 0)
        g.sync()
        dt_iso = one_pass()
        prof_iso = g.profile()
        if g.checksum() != cs:
            raise RuntimeError("the non-overlapped pass built another graph")
    g.close()
    torch.cuda.empty_cache()
    W = (2 * k + 63) // 64
    bpk = (0.375 if packed is not None else 1.0) * (READ_LEN + 1) / (READ_LEN - k + 1.0)
    kt = kernel_table(prof_iso if prof_iso is not None else prof, st.num_kmers_loaded, W, bpk)
    dom = max(kt, key=lambda n: kt[n]["total_ms"])
    alg = (ALG_BYTES_PER_KMER if W == 1 else 29.70) * st.num_kmers_loaded + 8.0 * W * st.num_kmers_novel
    out = {"value": st.num_kmers_loaded / dt, "unit": "k-mers/s", "ms_per_step": 1e3 * dt / len(batches), "steps": len(batches),
           "kmers_inserted": int(st.num_kmers_loaded), "distinct_kmers": int(st.num_kmers_novel),
           "roofline_frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4),
           "table_passes": ist["flushes"], "fallback_inserts": ist["fallback_inserts"]}
    if prof_iso is not None:
        out.update({"dominant_kernel": dom, "dominant_frac": kt[dom]["frac"], "kernels": kt,
                    "kernels_note": "ISOLATED durations: a second pass over the same input with the flush overlap off (one kernel at a time, HIP events around "
                                    "every launch, same graph checksum); dominant_frac = that kernel's own design bytes over its isolated time, against 8 TB/s",
                    "isolated_pass_ms_per_step": round(1e3 * dt_iso / len(batches), 3),
                    "spans_overlapped": {n: {"launches": c, "total_ms": round(t, 3)} for n, (c, t) in prof.items()}})
    else:
        out.update({"spans_overlapped": {n: {"launches": c, "total_ms": round(t, 3)} for n, (c, t) in prof.items()},
                    "kernels_note": "HIP-event SPANS of the timed pass: with the flush overlap on (the default) the spans of k_tuples_bin and k_lds_insert contain "
                                    "each other, so no per-kernel fraction is derived from them (isolated figures: the headline, C4_k63, C2_stress, hashtest)"})
    if cs is not None and checksum:
        out["graph_checksum"], out["nodes"] = "%016x" % cs[0], int(cs[1])
    return out


def ceilings(mcx, table_bytes):
    """the streaming and the random-RMW ceiling of THIS device, measured in this run (mcx_ubench_*)"""
    out = {}
    try:
        st = mcx.ubench_stream(8 << 30)
        out.update(measured_copy_gbs=round(st["copy"], 1), measured_read_gbs=round(st["read"], 1), measured_write_gbs=round(st["write"], 1))
        rw = mcx.ubench_random_rmw(table_bytes, 1 << 29)
        out.update(measured_random_rmw_per_s=round(rw["rmw"]), measured_random_load16_per_s=round(rw["load16"]),
                   measured_random_load_rmw_per_s=round(rw["load_rmw"]), random_working_set_bytes=int(table_bytes))
    except Exception as e:
        out["ceilings_error"] = str(e)[:200]
    return out


def hashtest(mcx, device, table_slots, nkeys=800_000_000):
    """The reference's own published benchmark: `mccortex31 hashtest -k 31 -n 1G 800000000`
    = hash_table_find_or_insert of the integer keys 0 .. N-1 (BinaryKmer b[0] = i; src/commands/ctx_exp_hashtest.c:40-69)
    into 1,073,741,824 slots (results/hash_table_benchmark/benchmark-tables.sh:53): 136.2 s = 5.9 M inserts/s on
    the reference's 2015 Xeon box (results20150409thurs.linux.txt:10; whole process, table calloc included).
    Here: the same keys, resident in HBM, through mcx_graph_insert_tuples_dev (full keys + an empty edge byte ->
    region bins -> split -> LDS insert; every insert also counts coverage, which the reference's loop does not)."""
    import torch
    keys = torch.arange(nkeys, dtype=torch.int64, device=device)
    edges = torch.zeros(nkeys, dtype=torch.uint8, device=device)
    g = mcx.Graph(K, 1, table_slots)
    g.insert_tuples_dev(0, keys[:65536], edges[:65536], 65536)
    g.sync(); g.reset(); g.sync()

    def one_pass():
        g.configure("profile", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chunk = 100_000_000
        for lo in range(0, nkeys, chunk):
            n = min(chunk, nkeys - lo)
            g.insert_tuples_dev(0, keys[lo:lo + n], edges[lo:lo + n], n)
        g.sync()
        return time.perf_counter() - t0

    dt = one_pass()
    nk = g.nkmers
    spans = g.profile()
    nkc, sc = g.kmer_covg()
    g.reset(); g.configure("flush_overlap", 0); g.sync()   # isolated per-kernel durations: a second pass, one kernel at a time
    dt_iso = one_pass()
    prof = g.profile()
    g.close()
    del keys, edges
    torch.cuda.empty_cache()
    alg = 24.0 * nkeys  # key in (8) + key slot read (8) + key write (8): every key is new
    return {"value": nkeys / dt, "unit": "inserts/s", "seconds": dt, "keys": nkeys, "table_slots": table_slots,
            "distinct_keys_in_table": int(nk), "all_inserted_once": bool(nk == nkeys and int(sc[0]) == nkeys),
            "roofline_frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes_per_insert": 24.0,
            "reference_published": {"seconds": 136.2, "inserts_per_s": 5.9e6, "threads": "-t 0 (single-thread code; its -t 1/2/4 runs are slower: 216 / 325 / 404 s)",
                                    "hardware": "8x 8-core Xeon 2.70 GHz, 2015 (other hardware: not a same-run comparison)",
                                    "source": "results/hash_table_benchmark/results20150409thurs.linux.txt:10"},
            "vs_reference_published": nkeys / dt / 5.9e6,
            "kernels": {n: {"launches": c, "total_ms": round(t, 3)} for n, (c, t) in prof.items()},
            "kernels_note": "ISOLATED durations (second pass, flush overlap off); spans of the timed pass: spans_overlapped",
            "isolated_pass_seconds": round(dt_iso, 4),
            "spans_overlapped": {n: {"launches": c, "total_ms": round(t, 3)} for n, (c, t) in spans.items()},
            "what": "mccortex31 hashtest -k 31 -n 1G 800000000: integer keys 0..N-1, device-resident, mcx_graph_insert_tuples_dev in chunks of 100 M + sync"}


def c2_stress(mcx, device, nsteps, batch_reads, slots=1 << 33):
    """50 M x 150 bp iid random reads (every k-mer novel: insert-bound worst case), a table of
    2^33 slots (128 GiB of records) so that the 6 G distinct k-mers fit at load 0.7; one flush."""
    import torch
    steps = [make_batch_iid(batch_reads, seed=7000 + i, device=device) for i in range(nsteps)]
    try:
        # (flush size: the stream positions of all steps, as for C4 -- a device-resident launch is booked with the positions
        # it covers until it has settled -- so that the build really makes ONE table pass: 2 M sub-table visits, not 4 M)
        r = run_config(mcx, steps, K, 1, [0] * nsteps, slots, nsteps * batch_reads * (READ_LEN + 1) + (1 << 26), isolated=True)
    finally:
        del steps
        torch.cuda.empty_cache()
    r["workload"] = ("C2-stress: k=31, 1 colour, %d iid random reads x %d bp per step (every k-mer novel), table %d slots (%d GiB), 1 GPU, one flush"
                     % (batch_reads, READ_LEN, slots, slots * 16 >> 30))
    return r


def write_fastq(path, batches, sample_path=None, sample_reads=500_000):
    """device-resident read batches -> a 4-line FASTQ file (constant quality 'I')"""
    import torch
    B = batches[0].numel() // (READ_LEN + 1)
    hdr = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device=batches[0].device)
    mid = torch.tensor(list(b"+\n"), dtype=torch.uint8, device=batches[0].device)
    with open(path, "wb") as f:
        for i, b in enumerate(batches):
            rec = torch.empty((B, 3 + (READ_LEN + 1) + 2 + (READ_LEN + 1)), dtype=torch.uint8, device=b.device)
            rec[:, :3] = hdr
            rec[:, 3:3 + READ_LEN + 1] = b.reshape(B, READ_LEN + 1)
            rec[:, 3 + READ_LEN + 1:3 + READ_LEN + 3] = mid
            rec[:, 3 + READ_LEN + 3:-1] = ord("I")
            rec[:, -1] = ord("\n")
            a = rec.cpu().numpy()
            a.tofile(f)
            if i == 0 and sample_path:
                a[:sample_reads].tofile(sample_path)
            del rec, a


def extras(mcx, batches, packed, nsteps, table_slots, oracle_steps=0, full_e2e=True):
    """host-fed, end-to-end, default flush size and the other BASELINE configs (rank 0, N = 1)"""
    import subprocess
    import tempfile
    import torch
    out = {}
    B = batches[0].numel() // (READ_LEN + 1)
    steps = batches[:nsteps]
    pk = packed[:nsteps] if packed is not None else None
    device = batches[0].device
    # the GPU graphs the oracle leg (cpu_baseline.full_size) and the full-size CLI run (e2e_full) are compared with
    if oracle_steps:
        r = run_config(mcx, batches[:oracle_steps], K, 1, [0] * oracle_steps, table_slots, DEFER_TUPLES, None, checksum=True)
        out["_gpu_oracle_steps"] = {"steps": oracle_steps, "graph_checksum": r["graph_checksum"], "nodes": r["nodes"], "kmers": r["kmers_inserted"]}
    n_c2 = 50_000_000 // B if B and 50_000_000 % B == 0 else 0  # steps of the whole 50 M-read set
    if full_e2e and n_c2 and nsteps >= n_c2:
        r = run_config(mcx, batches[:n_c2], K, 1, [0] * n_c2, table_slots, DEFER_TUPLES, None, checksum=True)
        out["_gpu_c2"] = {"steps": n_c2, "graph_checksum": r["graph_checksum"], "nodes": r["nodes"], "kmers": r["kmers_inserted"]}
    # (d) the library's own flush size (64 occurrences per slot within 30 % of the free HBM) instead of the bench's
    r = run_config(mcx, steps, K, 1, [0] * len(steps), table_slots, 0, pk)
    r["what"] = "as `value`, but with the library's default flush size instead of %d occurrences" % DEFER_TUPLES
    out["default_defer"] = r
    if pk is not None:  # the same build from the ASCII form of the stream (what `value` is by default)
        r = run_config(mcx, steps, K, 1, [0] * len(steps), table_slots, HEADLINE_DEFER, None)
        r["what"] = "as `value`, but the resident stream is ASCII (1 byte per position; the kernel encodes it in its tile prologue)"
        out["ascii_resident"] = r
    else:               # ... and from the packed form the host entry stages (packing outside the clock: round 2's `value`)
        pk2 = pack_batches(mcx, steps)
        r = run_config(mcx, steps, K, 1, [0] * len(steps), table_slots, HEADLINE_DEFER, pk2)
        r["what"] = "as `value`, but the resident stream is already packed (2-bit codes + invalid flags, 3 bits per position): the packing is OUTSIDE this clock"
        out["packed_resident"] = r
        del pk2
        torch.cuda.empty_cache()
    # (e) C4: k = 63 (two-word keys); C5-like: 4 colours on one GPU
    # (flush size: the k-mers of all steps + the launches in flight, as for `value` -- the library books a stream launch with
    # the positions it covers, 1.7x the k-mers it yields at k = 63, and takes the excess off the books as launches settle --
    # so that the build makes ONE table pass like C2's: 9.1 G two-word tuples = 154 GB of bins at 20 steps)
    r = run_config(mcx, steps, 63, 1, [0] * len(steps), table_slots, max(5_000_000_000, int(len(steps) * B * (READ_LEN - 63 + 1) * 1.035) + B * (READ_LEN + 1)), pk, isolated=True)
    r["workload"] = "C4: k=63 (2-word BinaryKmer), 1 colour, %d reads x %d bp per step, table %d slots, 1 GPU" % (B, READ_LEN, table_slots)
    out["other_configs"] = {"C4_k63": r}
    # (4 colours: the L1 workspace is a pool of bin sets shared by the colours, sized here to hold all 20 steps,
    # so that the build makes one table pass per colour however the samples are ordered)
    c5_defer = max(6_000_000_000, len(steps) * B * (READ_LEN + 1) + (1 << 28))  # (counted in start positions: a colour's bookings are spread over its bin sets)
    cols = [min(3, 4 * i // max(1, len(steps))) for i in range(len(steps))]
    r = run_config(mcx, steps, K, 4, cols, table_slots, c5_defer, pk)
    r["workload"] = "C5-like: k=31, 4 colours (steps dealt out to samples %s), table %d slots, ONE GPU" % (cols, table_slots)
    out["other_configs"]["C5_like_4_colours_1gpu"] = r
    cols = [i % 4 for i in range(len(steps))]  # a population build alternates samples: colour switches at every step
    r = run_config(mcx, steps, K, 4, cols, table_slots, c5_defer, pk)
    r["workload"] = "C5-like, colours interleaved: k=31, 4 colours (steps dealt out to samples %s), table %d slots, ONE GPU" % (cols, table_slots)
    out["other_configs"]["C5_like_interleaved_colours_1gpu"] = r
    # C5 as the N > 1 run builds it (config.C5 there: the timed reads dealt out to 4 samples by their place in the input):
    # the same coloured graph on one GPU -- its checksum is what N1_C5_CHECKSUMS holds for the N > 1 runs to compare with
    try:
        per_col = colour_slices(steps, 0, B, 4)
        g = mcx.Graph(K, 4, table_slots)
        g.configure("defer_tuples", c5_defer)
        g.add_stream_dev(0, steps[0][:1024 * (READ_LEN + 1)], 1024 * (READ_LEN + 1))
        g.sync(); g.reset(); g.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c, lst in enumerate(per_col):
            for t in lst:
                g.add_stream_dev(c, t, t.numel())
        g.sync()
        dt = time.perf_counter() - t0
        st, ist = g.device_stats(), g.insert_stats()
        cs, nodes = g.checksum()
        g.close()
        torch.cuda.empty_cache()
        out["other_configs"]["C5_reference_1gpu"] = {
            "value": st.num_kmers_loaded / dt, "unit": "k-mers/s", "ms_per_step": 1e3 * dt / len(steps), "steps": len(steps), "kmers_inserted": int(st.num_kmers_loaded),
            "distinct_kmers": int(nodes), "graph_checksum": "%016x" % cs, "table_passes": ist["flushes"], "fallback_inserts": ist["fallback_inserts"],
            "matches_known": (("%016x" % cs) == N1_C5_CHECKSUMS[len(steps)]) if (len(steps) in N1_C5_CHECKSUMS and B == BATCH_READS) else None,
            "workload": "C5: k=31, 4 colours, the %d x %d timed reads dealt out to 4 samples in input order (%d reads each), table %d slots, ONE GPU; the graph "
                        "`bench.py --gpus N` builds as config.C5" % (len(steps), B, len(steps) * B // 4, table_slots)}
    except Exception as e:
        out["other_configs"]["C5_reference_1gpu"] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()
    # the reference's own published table benchmark, and the insert-bound worst case of SURVEY 8(d)
    for name, fn in (("hashtest", lambda: hashtest(mcx, device, table_slots)),
                     ("C2_stress", lambda: c2_stress(mcx, device, min(10, nsteps), B))):
        try:
            out["other_configs"][name] = fn()
        except Exception as e:
            out["other_configs"][name] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
    # (f) the multi-GPU table of the C ABI (mcx_graph_create_multi) with BOTH shards on this one GPU:
    # sender kernel -> peer copy (device-local here) -> owner split -> LDS insert; a check of the
    # code path and of its overheads, not a scaling figure
    for nsh, defer_sh in ((2, 3_000_000_000), (8, 1_000_000_000)):
      try:
        g = mcx.Graph(K, 1, table_slots, devices=[0] * nsh)
        g.configure("defer_tuples", defer_sh)
        if pk is not None:
            g.add_packed_dev(0, pk[0][0][:4096], pk[0][1][:4096], 65536)
        else:
            g.add_stream_dev(0, steps[0][:1024 * (READ_LEN + 1)], 1024 * (READ_LEN + 1))
        g.sync(); g.reset(); g.sync()
        g.configure("profile", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, b in enumerate(steps):
            if pk is not None:
                g.add_packed_dev(0, *pk[i])
            else:
                g.add_stream_dev(0, b, b.numel())
        g.sync()
        dt = time.perf_counter() - t0
        st = g.device_stats()
        ist = g.insert_stats()
        # HIP-event spans of every shard's kernels, summed over the shards (all of which share the ONE device here, so
        # the spans of different shards overlap each other: a breakdown of where the launches go, not a time budget)
        stage_of = {"k_stream_superk": "sender", "k_stream_bin": "sender", "k_superk_bin": "owner", "k_tuples_bin": "split", "k_lds_insert": "insert"}
        stages = {}
        for kn, (c, t) in g.profile().items():
            e = stages.setdefault(stage_of.get(kn.split("@")[0], kn.split("@")[0]), {"launches": 0, "span_ms_summed_over_shards": 0.0})
            e["launches"] += c
            e["span_ms_summed_over_shards"] = round(e["span_ms_summed_over_shards"] + t, 2)
        cs, nk = g.checksum()
        g.close()
        torch.cuda.empty_cache()
        out["inprocess_%d_shards_1gpu" % nsh] = {"value": st.num_kmers_loaded / dt, "unit": "k-mers/s", "ms_per_step": 1e3 * dt / len(steps),
                                          "distinct_kmers": int(st.num_kmers_novel), "graph_checksum": "%016x" % cs,
                                          "stages": stages, "table_passes_all_shards": ist["flushes"], "fallback_inserts": ist["fallback_inserts"], "spilled": ist["spilled"],
                                          "what": "mcx_graph_create_multi with device 0 named %d times: %d shards on ONE GPU, exchange v3 (super-k-mer records, "
                                                  "minimizer owners), the filled parts copied by a kernel through peer-mapped pointers (device-local here); a "
                                                  "measure of the path's overhead, not of scaling; the graph checksum must equal config.graph_checksum" % (nsh, nsh)}
      except Exception as e:
        out["inprocess_%d_shards_1gpu" % nsh] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()
    # (b) the same reads handed over in HOST memory through mcx_graph_add_reads (pinned buffers):
    # staging, PCIe and the kernels inside the clock
    nh = len(steps)
    hb = []
    for b in steps[:nh]:
        t = torch.empty((B, READ_LEN), dtype=torch.uint8).pin_memory()
        t.copy_(b.reshape(B, READ_LEN + 1)[:, :READ_LEN])
        hb.append(t.numpy().reshape(-1))
    offs = np.arange(B + 1, dtype=np.uint64) * READ_LEN
    g = mcx.Graph(K, 1, table_slots)
    g.add_reads(0, hb[0][:READ_LEN * 1000], offs[:1001]); g.sync(); g.reset(); g.sync()
    t0 = time.perf_counter()
    for h in hb:
        g.add_reads(0, h, offs)
    g.sync()
    dt = time.perf_counter() - t0
    st = g.device_stats()
    g.close()
    torch.cuda.empty_cache()
    out["host_fed"] = {"value": st.num_kmers_loaded / dt, "unit": "k-mers/s", "steps": nh, "ms_per_step": 1e3 * dt / nh,
                       "host_gb_per_s": sum(h.size for h in hb) / dt / 1e9,
                       "what": "reads as concatenated ASCII bases + offsets in pinned host memory -> mcx_graph_add_reads "
                               "(staging threads, H2D and kernels inside the clock), %d x %d reads" % (nh, B)}
    # (c) end to end: mccortex31 build --sort on a FASTQ file of the same reads
    exe = os.path.join(ROOT, "mccortex_amd", "bin", "mccortex31")
    tmp = tempfile.mkdtemp(prefix="mcx_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    fq, fq_small, ctx = os.path.join(tmp, "reads.fq"), os.path.join(tmp, "sample.fq"), os.path.join(tmp, "out.ctx")
    try:
        ne = min(2, len(steps))
        write_fastq(fq, steps[:ne], fq_small)
        os.sync()   # (input preparation ends here: the file's dirty pages are on disk before the command is timed)
        nthreads = min(32, os.cpu_count() or 1)
        cmd = [exe, "build", "-f", "-k", str(K), "-n", str(table_slots), "-m", "%dG" % (table_slots * 21 // (1 << 30) + 2), "-t", str(nthreads),
               "--sort", "--sample", "bench", "--seq", fq, ctx]
        best = None
        runs, recs = [], []
        for _ in range(3):  # later runs: file in the page cache, HIP kernels' code objects loaded before
            if os.path.exists(ctx):
                os.unlink(ctx)  # (overwriting 3 GB of dirty page cache is not part of the command)
            time.sleep(2.0)  # (this process, then the previous run, has just freed tens of GB of device memory: the driver scrubs it)
            t0, w0 = time.perf_counter(), time.time()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MCX_TIMING="1"))
            dt, w1 = time.perf_counter() - t0, time.time()
            if p.returncode != 0:
                raise RuntimeError("mccortex31 build failed: " + p.stderr.decode(errors="replace")[-400:])
            runs.append(round(dt, 3))
            recs.append((dt, p.stderr.decode(errors="replace"), w0, w1))
        best = sorted(recs, key=lambda r: r[0])[len(recs) // 2]  # the MEDIAN run (all three are listed in runs_s)
        kmers = ne * B * (READ_LEN - K + 1)  # upper bound; the exact figure: reads with an N lose a few
        for line in best[1].splitlines():
            if "kmers" in line and "Loaded" in line:
                pass
        stages = [ln.strip() for ln in best[1].splitlines() if ln.startswith("[timing]")]
        ep = {ln.split()[1]: float(ln.split()[2]) for ln in stages if "epoch_" in ln}
        outside = {}
        if "epoch_main_entry" in ep and "epoch_main_exit" in ep:  # what the command itself does not see
            outside = {"spawn_to_main_s": round(ep["epoch_main_entry"] - best[2], 3), "main_s": round(ep["epoch_main_exit"] - ep["epoch_main_entry"], 3),
                       "exit_to_reaped_s": round(best[3] - ep["epoch_main_exit"], 3)}
        out["e2e"] = {"value": kmers / best[0], "unit": "k-mers/s (upper bound on k-mers: %d per read)" % (READ_LEN - K + 1),
                      "seconds": best[0], "fastq_bytes": os.path.getsize(fq), "ctx_bytes": os.path.getsize(ctx),
                      "command": "mccortex31 build -k %d -n %d -m %dG -t %d --sort --seq <%d-read FASTQ> out.ctx" % (K, table_slots, table_slots * 21 // (1 << 30) + 2, nthreads, ne * B),
                      "what": "wall clock of the whole process (start, HIP init, parse, build, device sort, .ctx write), MEDIAN of 3 runs (all in runs_s; stages are the median run's)",
                      "runs_s": runs,
                      "process": outside, "stages": [x for x in stages if "epoch_" not in x][-14:]}
        out["_fastq_sample"] = fq_small
        out["_tmpdir"] = tmp
    except Exception as e:  # the extras must never cost the headline its line
        out["e2e"] = {"error": str(e)[:300]}
        out["_tmpdir"] = tmp
    # (c') BASELINE config 2 end to end at its full size: the CLI on the whole 50 M x 150 bp FASTQ (15 GB), once;
    # the body of the .ctx it writes must have the checksum of the device-resident build of the same reads
    if "_gpu_c2" in out:
        try:
            import shutil
            n_c2 = out["_gpu_c2"]["steps"]
            need = n_c2 * B * (2 * READ_LEN + 7) + 13 * out["_gpu_c2"]["nodes"] + (2 << 30)
            if shutil.disk_usage(tmp).free < need:
                raise RuntimeError("not enough scratch space in %s for a %d GB FASTQ + .ctx" % (tmp, need >> 30))
            for f_ in (fq, ctx):
                if os.path.exists(f_):
                    os.unlink(f_)
            t0 = time.perf_counter()
            write_fastq(fq, batches[:n_c2])
            os.sync()   # (the 15 GB just written are dirty pages: without this their write-back competes with the command's own 4.4 GB of output)
            t_write = time.perf_counter() - t0
            nthreads = min(32, os.cpu_count() or 1)
            cmd = [exe, "build", "-f", "-k", str(K), "-n", str(table_slots), "-m", "%dG" % (table_slots * 21 // (1 << 30) + 2), "-t", str(nthreads),
                   "--sort", "--sample", "bench", "--seq", fq, ctx]
            time.sleep(2.0)
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MCX_TIMING="1"))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                raise RuntimeError("mccortex31 build failed: " + p.stderr.decode(errors="replace")[-400:])
            stages = [ln.strip() for ln in p.stderr.decode(errors="replace").splitlines() if ln.startswith("[timing]") and "epoch_" not in ln]
            fq_bytes, ctx_bytes = os.path.getsize(fq), os.path.getsize(ctx)
            os.unlink(fq)
            hdr = 6 + 16 + 4 + 8 + (4 + len("bench")) + 16 + (12 + 4 + 9) + 6   # .ctx v6 header, one colour named "bench"
            body = np.fromfile(ctx, dtype=np.uint8, offset=hdr)
            assert body.size % 13 == 0, "unexpected .ctx size"
            cs = mcx.records_checksum(body, K, 1)
            nrec = body.size // 13
            del body
            os.unlink(ctx)
            kmers = out["_gpu_c2"]["kmers"]
            out["e2e_full"] = {"value": kmers / dt, "unit": "k-mers/s", "seconds": dt, "reads": n_c2 * B, "kmers": kmers, "fastq_bytes": fq_bytes, "ctx_bytes": ctx_bytes,
                               "ctx_records": int(nrec), "ctx_body_checksum": "%016x" % cs,
                               "checksum_matches_device_resident_build": ("%016x" % cs) == out["_gpu_c2"]["graph_checksum"] and nrec == out["_gpu_c2"]["nodes"],
                               "command": "mccortex31 build -k %d -n %d -m %dG -t %d --sort --seq <%d-read FASTQ> out.ctx" % (K, table_slots, table_slots * 21 // (1 << 30) + 2, nthreads, n_c2 * B),
                               "what": "BASELINE config 2 end to end at full size, ONE run: process start, HIP init, parse of the %.1f GB FASTQ (page cache), build, device sort, "
                                       ".ctx write; the written body's order-independent checksum against the device-resident build of the same %d steps" % (fq_bytes / 1e9, n_c2),
                               "fastq_write_s_outside_clock": round(t_write, 1), "stages": stages[-14:]}
        except Exception as e:
            out["e2e_full"] = {"error": str(e)[:300]}
    return out


def spawn_ranks(n):
    """re-exec this command under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous on a free
    port); stdout of the ranks is this process's stdout, so rank 0's JSON line is the one line printed"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def effective_cores():
    """CPU time the container grants per second of wall clock (cgroup quota / period), or None when unlimited"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] == "max":
                    return None
                return float(txt[0]) / float(txt[1])
            q = float(txt[0])
            if q <= 0:
                return None
            return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
        except (OSError, ValueError, IndexError):
            continue
    return None


def colour_slices(step_batches, first_read, B, ncols):
    """C5's samples: the job's timed reads in order (step 0's B reads, step 1's, ...) are dealt out to `ncols` samples,
    reads [c T / ncols, (c + 1) T / ncols) of the T = steps x B to sample c (4 x 12.5 M reads at 10 steps) -- a function of
    the read's place in the N = 1 input, so that every N builds the same coloured graph.  step_batches[i] holds reads
    [first_read, first_read + n_i) of step i (this rank's slice).  -> per colour, the list of its stream slices."""
    nsteps = len(step_batches)
    T = nsteps * B
    out = [[] for _ in range(ncols)]
    for i, b in enumerate(step_batches):
        rows = b.reshape(-1, READ_LEN + 1)
        g0 = i * B + first_read                      # place of this slice's first read in the whole input
        for c in range(ncols):
            lo, hi = max(T * c // ncols, g0), min(T * (c + 1) // ncols, g0 + rows.shape[0])
            if hi > lo:
                out[c].append(rows[lo - g0:hi - g0].reshape(-1))
    return out


def _deal_colours(step_batches, first_read, B, ncols, xmax, shard, dist, world, device):
    """colour_slices() cut into exchange steps of at most xmax bytes: [(colour, stream)], the same number of steps per
    colour on every rank (a rank with fewer reads of a sample passes shorter, possibly empty, steps)"""
    per_col = colour_slices(step_batches, first_read, B, ncols)
    streams = [torch.cat(v) if v else torch.zeros(0, dtype=torch.uint8, device=device) for v in per_col]
    most = torch.tensor([max(t.numel() for t in streams)], dtype=torch.int64, device=device)
    if world > 1:
        shard.all_reduce(most, op=dist.ReduceOp.MAX)
    reads_per_x = max(1, xmax // (READ_LEN + 1))
    nx = max(1, -(-(int(most.item()) // (READ_LEN + 1)) // reads_per_x))
    out = []
    for c, t in enumerate(streams):
        rows = t.reshape(-1, READ_LEN + 1)
        n = rows.shape[0]
        for j in range(nx):
            out.append((c, rows[n * j // nx:n * (j + 1) // nx].reshape(-1)))
    return out


def sharded_pass(mcx, shard, dist, args, device, local_rank, rank, world, fmt, ncols, x_timed, x_warm, xmax, slots_per_gpu, nsteps):
    """ONE timed pass of the N > 1 path -- sender kernel -> all-to-all -> owner kernels (mccortex_amd/shard.py:
    ShardedInserter), fresh graph -- in exchange format `fmt` ("v3": minimizer-owned super-k-mer records, ordinary
    per-rank tables; "v2": packed tuples, table sharded by quotient-hash prefix).  x_timed / x_warm: [(colour, stream)].
    Every rank returns the job-wide record (totals all-reduced, per-rank stage table all-gathered)."""
    use_v3 = fmt == "v3"
    if use_v3:
        graph = mcx.Graph(K, ncols, slots_per_gpu, device=local_rank)
    else:
        graph = mcx.Graph(K, ncols, slots_per_gpu, device=local_rank, nparts=world, part=rank)
    graph.configure("defer_tuples", args.defer_tuples)
    probe = x_timed[0][1] if x_timed[0][1].numel() >= 1024 * (READ_LEN + 1) else max((x for _, x in x_timed), key=lambda t: t.numel())
    nprobe = min(probe.numel(), 1024 * (READ_LEN + 1))
    graph.add_stream_dev(0, probe[:nprobe], nprobe)   # the bin workspace is allocated on first use: outside the timed region
    graph.sync()
    graph.reset()
    inserter = shard.ShardedInserter(graph, world, device, xmax, use_v3, max_tuples=xmax // (READ_LEN + 1) * (READ_LEN - K + 1))

    def run(xs):
        i = 0
        while i < len(xs):   # consecutive exchange steps of one colour are one insert() call (double buffered inside)
            j = i
            while j < len(xs) and xs[j][0] == xs[i][0]:
                j += 1
            inserter.insert(xs[i][0], [(x, x.numel()) for _, x in xs[i:j]])
            i = j

    def fence():
        torch.cuda.synchronize()
        graph.sync()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(x_warm)
    fence()
    graph.reset()
    fence()
    inserter.reset_stats()
    graph.configure("profile", 1)  # HIP events around every kernel launch on the handle's stream
    t0 = time.perf_counter()
    run(x_timed)
    fence()
    dt_local = time.perf_counter() - t0

    st = graph.device_stats()
    ist = graph.insert_stats()
    prof = graph.profile()
    cs_local, nodes_local = graph.checksum()
    kmers_local = st.num_kmers_loaded
    ident = torch.tensor([cs_local & 0xFFFFFFFF, cs_local >> 32, nodes_local, kmers_local, ist["flushes"], ist["fallback_inserts"], ist["foreign_inserts"]],
                         dtype=torch.int64, device=device)
    tmax = torch.tensor([dt_local, float(ist["flushes"])], dtype=torch.float64, device=device)
    if world > 1:
        shard.all_reduce(ident, op=dist.ReduceOp.SUM)   # 32-bit halves: the sums cannot overflow
        shard.all_reduce(tmax, op=dist.ReduceOp.MAX)
    iv = [int(v) for v in ident.tolist()]
    cs_total = (iv[0] + (iv[1] << 32)) & 0xFFFFFFFFFFFFFFFF
    dt = float(tmax[0].item())
    stage_of = {"k_stream_superk": "sender", "k_stream_bin": "sender", "k_superk_bin": "owner", "k_tuples_bin": "split",
                "k_lds_insert": "insert", "k_insert_tuples": "owner_overflow"}
    mine = {"rank": rank, "device": local_rank, "wall_s": round(dt_local, 5),
            "kmers_kmerised": int(kmers_local), "distinct_kmers_owned": int(nodes_local), "table_passes": ist["flushes"],
            "stage_ms": {}, "exchange_ms": round(inserter.stats["exchange_ms"], 3), "exchange_steps": inserter.stats["steps"],
            "link_bytes_sent": inserter.stats["link_bytes_sent"], "stream_bytes": inserter.stats["stream_bytes"]}
    for kn, (c, t) in prof.items():
        e = mine["stage_ms"].setdefault(stage_of.get(kn, kn), {"kernel": kn, "launches": 0, "total_ms": 0.0})
        e["launches"] += c
        e["total_ms"] = round(e["total_ms"] + t, 3)
    mine["link_bytes_per_occurrence"] = round(mine["link_bytes_sent"] / max(1, kmers_local), 3)
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]
    graph.close()
    del inserter, graph
    torch.cuda.empty_cache()
    kmers_total = iv[3]
    link = sum(r["link_bytes_sent"] for r in per_rank)
    return {"exchange_format": fmt, "colours": ncols, "value": kmers_total / dt, "unit": "k-mers/s", "seconds": dt, "ms_per_step": 1e3 * dt / nsteps,
            "kmers_inserted": kmers_total, "distinct_kmers_total": iv[2], "graph_checksum": "%016x" % cs_total,
            "exchange_steps": per_rank[0]["exchange_steps"], "link_bytes_total": link, "link_bytes_per_occurrence": round(link / max(1, kmers_total), 3),
            "exchange_ms_max": max(r["exchange_ms"] for r in per_rank),
            "stage_ms_max": {sname: round(max(r["stage_ms"].get(sname, {}).get("total_ms", 0.0) for r in per_rank), 3)
                             for sname in sorted({k_ for r in per_rank for k_ in r["stage_ms"]})},
            "table_passes_max": int(tmax[1].item()), "fallback_inserts_total": iv[5], "foreign_inserts_total": iv[6],
            "sharding": ("minimizer-owned super-k-mers x%d, all-to-all" if use_v3 else "hash-prefix x%d, all-to-all") % world,
            "per_rank": per_rank}


def inprocess_multi(args):
    """The C ABI's ONE-process table over devices 0..N-1 (mcx_graph_create_multi: the driver `mccortex31 build -D 0,1,..`
    runs; peer-mapped copy kernel instead of RCCL) on the N = 1 reads: every device holds its 1/N of every step's
    reads and k-merises them as the sender, the owners insert.  Run by the N > 1 bench as a subprocess (a fault on a
    path that has never met two GPUs must not cost the RCCL figures their line)."""
    import __graft_entry__
    __graft_entry__.build()
    import mccortex_amd as mcx
    n, B, nsteps, nwarm = args.inprocess_multi, args.batch_reads, args.steps, args.warmup
    dev0 = torch.device("cuda", 0)
    genome = make_genome(args.genome, dev0, seed=42)
    per_dev = [[] for _ in range(n)]
    for i in range(nsteps + nwarm):
        full = make_batch(genome, B, seed=1000 + i, device=dev0, err_rate=args.err).reshape(B, READ_LEN + 1)
        for d in range(n):
            per_dev[d].append(full[B * d // n:B * (d + 1) // n].reshape(-1).to(torch.device("cuda", d)))
        del full
    del genome
    for d in range(n):
        torch.cuda.synchronize(d)
    torch.cuda.empty_cache()
    g = mcx.Graph(K, 1, args.table_slots, devices=list(range(n)))
    g.configure("defer_tuples", max(1 << 28, args.defer_tuples // n))

    def run(idx):
        for i in idx:
            for d in range(n):
                g.add_stream_dev(0, per_dev[d][i], per_dev[d][i].numel())
        g.sync()

    run(range(nsteps, nsteps + nwarm))
    g.reset(); g.sync()
    g.configure("profile", 1)
    t0 = time.perf_counter()
    run(range(nsteps))
    dt = time.perf_counter() - t0
    st, ist = g.device_stats(), g.insert_stats()
    cs, nodes = g.checksum()
    stage_of = {"k_stream_superk": "sender", "k_stream_bin": "sender", "k_superk_bin": "owner", "k_tuples_bin": "split", "k_lds_insert": "insert"}
    stages = {}
    for kn, (c, t) in g.profile().items():
        name, _, shard_ix = kn.partition("@")
        e = stages.setdefault(stage_of.get(name, name), {})
        e[shard_ix or "0"] = round(e.get(shard_ix or "0", 0.0) + t, 2)
    g.close()
    return {"value": st.num_kmers_loaded / dt, "unit": "k-mers/s", "n_gpus": n, "seconds": dt, "ms_per_step": 1e3 * dt / nsteps, "steps": nsteps,
            "kmers_inserted": int(st.num_kmers_loaded), "distinct_kmers_total": int(nodes), "graph_checksum": "%016x" % cs,
            "table_passes_all_shards": ist["flushes"], "fallback_inserts": ist["fallback_inserts"], "spilled": ist["spilled"],
            "stage_span_ms_by_shard": stages,
            "what": "mcx_graph_create_multi over devices 0..%d in ONE process (one host thread issues every launch; exchange over peer-mapped "
                    "pointers, k_copy_filled), the N = 1 reads dealt out to the devices, %d slots in total" % (n - 1, args.table_slots)}


def sharded_main(mcx, shard, dist, args, device, local_rank, rank, world, force_shard):
    """N > 1 (one process per GPU; also N = 1 under MCX_BENCH_FORCE_SHARD=1): the exchange path in BOTH formats on one
    colour, then config C5 (4 colours) in the faster one, then the C ABI's one-process driver as a subprocess.
    `value` is the better one-colour figure; everything else sits in `config`."""
    B, nsteps, nwarm = args.batch_reads, args.steps, args.warmup
    strong = args.scaling == "strong" and not args.iid
    if args.iid:
        batches = [make_batch_iid(B, seed=1000 * (rank + 1) + i, device=device) for i in range(nsteps + nwarm)]
    elif strong:
        # C3: the reads of the N = 1 run.  Step i is the batch N = 1 uses for step i (same seed);
        # this rank takes reads [rank B / N, (rank + 1) B / N) of it.
        genome = make_genome(args.genome, device, seed=42)
        r_lo, r_hi = B * rank // world, B * (rank + 1) // world
        batches = []
        for i in range(nsteps + nwarm):
            full = make_batch(genome, B, seed=1000 + i, device=device, err_rate=args.err)
            batches.append(full.reshape(B, READ_LEN + 1)[r_lo:r_hi].reshape(-1).clone())
            del full
        del genome
    else:
        genome = make_genome(args.genome * world, device, seed=42)
        batches = [make_batch(genome, B, seed=1000 * (rank + 1) + i, device=device, err_rate=args.err) for i in range(nsteps + nwarm)]
        del genome
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    slots_per_gpu = max(1 << 20, args.table_slots // world) if strong else args.table_slots

    def exchange_steps(idx):
        """the streams one all-to-all step moves: a rank's own batch (weak), or -- strong, where a rank's share of a
        step is B / N reads -- its shares of N consecutive steps in one piece, so that an exchange step carries as
        many reads as at N = 1"""
        idx = list(idx)
        if not strong:
            return [batches[i] for i in idx]
        return [torch.cat([batches[i] for i in idx[j:j + world]]) for j in range(0, len(idx), world)]

    x_timed, x_warm = exchange_steps(range(nsteps)), exchange_steps(range(nsteps, nsteps + nwarm))
    per_step_max = (-(-B // world)) if strong else B          # reads of one step on the fullest rank
    xmax = (per_step_max * (world if strong else 1) + 1) * (READ_LEN + 1)   # same on every rank

    formats = ["v3", "v2"] if mcx.superk_supported(K) else ["v2"]
    if os.environ.get("MCX_EXCHANGE") in formats:
        formats = [os.environ["MCX_EXCHANGE"]]   # (tests, experiments: one format only)
    one = {}
    for fmt in formats:
        one[fmt] = sharded_pass(mcx, shard, dist, args, device, local_rank, rank, world, fmt, 1,
                                [(0, x) for x in x_timed], [(0, x) for x in x_warm], xmax, slots_per_gpu, nsteps)
    best = max(one, key=lambda f: one[f]["value"])
    c5 = None
    if not args.iid and os.environ.get("MCX_BENCH_C5", "1") != "0":
        c5_timed = _deal_colours(batches[:nsteps], (B * rank // world) if strong else 0, B, 4, xmax, shard, dist, world, device)
        c5_warm = [(i % 4, x) for i, x in enumerate(x_warm)]
        c5 = sharded_pass(mcx, shard, dist, args, device, local_rank, rank, world, best, 4, c5_timed, c5_warm, xmax, slots_per_gpu, nsteps)
        del c5_timed, c5_warm
    del x_timed, x_warm, batches
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    r = one[best]
    scaling = "strong" if strong else "weak"
    if strong:
        shape = ("C3: the N=1 reads (%d reads x %d bp per step from a %d Mbp random genome) dealt out to %d GPUs, %d reads per step per GPU, "
                 "table %d slots in total = %d per GPU" % (B, READ_LEN, args.genome // 1_000_000, world, B // world, args.table_slots, slots_per_gpu))
    else:
        shape = ("%d reads x %d bp per step per GPU from a %d Mbp random genome, table %d slots per GPU" % (B, READ_LEN, args.genome * world // 1_000_000, args.table_slots))
    transport = shard.dist_backend() + (" (TEST transport: host-staged collectives, all ranks on one device)" if shard.dist_backend() != "nccl" else " (RCCL)")
    cfg = {"workload": ("C2-stress: k=31, 1 colour, %d iid random reads x %d bp per step per GPU, table %d slots per GPU" % (B, READ_LEN, args.table_slots)) if args.iid else
                       "C2: k=31, 1 colour, %s; input already resident in HBM as an ASCII byte stream (parse and H2D outside `value`)" % shape,
           "input": "device-resident ASCII stream", "kmer_size": K, "colours": 1,
           "reads_per_step_per_gpu": (B // world) if strong else B, "read_len": READ_LEN,
           "table_slots_per_gpu": slots_per_gpu, "table_slots_total": slots_per_gpu * world,
           "sharding": r["sharding"], "exchange_format": best, "transport": transport,
           "insert_path": "partition + LDS insert, %d occurrences per flush" % args.defer_tuples,
           "kmers_inserted": r["kmers_inserted"], "distinct_kmers_total": r["distinct_kmers_total"], "graph_checksum": r["graph_checksum"],
           "table_passes_max_over_ranks": r["table_passes_max"], "fallback_inserts_total": r["fallback_inserts_total"], "foreign_inserts_total": r["foreign_inserts_total"]}
    # the same reads must give the same graph whatever N and whatever the exchange format
    known = (strong or world == 1) and not args.iid and B == BATCH_READS and args.genome == GENOME_PER_GPU and args.err == 0.001 \
        and READ_LEN == 150 and nsteps in N1_CHECKSUMS
    if known:
        cfg["checksum_matches_n1"] = all(v["graph_checksum"] == N1_CHECKSUMS[nsteps] for v in one.values())
    cfg["formats_agree"] = len({(v["graph_checksum"], v["distinct_kmers_total"], v["kmers_inserted"]) for v in one.values()}) == 1
    slim = lambda v: {k_: v[k_] for k_ in v if k_ != "per_rank"}
    cfg["exchange_formats"] = {f: dict(slim(v), per_rank=v["per_rank"]) for f, v in one.items()}
    cfg["exchange_formats"]["what"] = ("the same steps timed once per exchange format (fresh tables each time; `value` is the faster one): per format the job-wide figure, "
                                       "max-over-ranks HIP-event spans per stage (sender = k-merise own reads into per-owner bins, owner = k-merise received super-k-mers "
                                       "into region bins, split = region -> sub-table bins, insert = LDS insert), the exchange's span on torch's stream, the bytes put "
                                       "on the links, and the per-rank table")
    if c5 is not None:
        c5["workload"] = ("C5: k=31, 4 colours, the timed reads dealt out to 4 samples (%d reads each job-wide), exchange format %s, table %d slots in total"
                          % (nsteps * B * (1 if strong else world) // 4, best, slots_per_gpu * world))
        # same reads as the one-colour pass: the union graph has the same nodes and the same number of occurrences
        c5["nodes_and_kmers_match_one_colour"] = (c5["distinct_kmers_total"] == r["distinct_kmers_total"] and c5["kmers_inserted"] == r["kmers_inserted"])
        if known and nsteps in N1_C5_CHECKSUMS:
            c5["checksum_matches_n1"] = c5["graph_checksum"] == N1_C5_CHECKSUMS[nsteps]
        cfg["C5"] = c5
    # the C ABI's one-process driver over the same devices (bounded subprocess; rank 0 only, the other ranks have
    # released their memory and wait at the closing barrier)
    if world > 1 and strong and shard.dist_backend() == "nccl" and os.environ.get("MCX_BENCH_INPROCESS", "1") != "0":
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--inprocess-multi", str(world), "--steps", str(nsteps), "--warmup", str(min(nwarm, 1)),
               "--batch-reads", str(B), "--table-slots", str(args.table_slots), "--genome", str(args.genome), "--err", str(args.err),
               "--defer-tuples", str(args.defer_tuples)]
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                   "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
        env["MCX_TIMING"] = "1"   # (peer matrix and self-test verdict on stderr)
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=420)
            lines = p.stdout.decode(errors="replace").strip().splitlines()
            if p.returncode == 0 and lines:
                rec = json.loads(lines[-1])
                rec["graph_matches_rccl_path"] = rec["graph_checksum"] == r["graph_checksum"] and rec["distinct_kmers_total"] == r["distinct_kmers_total"]
                rec["peer_selftest"] = [ln for ln in p.stderr.decode(errors="replace").splitlines() if "peer" in ln][-12:]
                cfg["inprocess_multi"] = rec
            else:
                cfg["inprocess_multi"] = {"error": "rc %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-400:])}
        except Exception as e:
            cfg["inprocess_multi"] = {"error": str(e)[:300]}
    # roofline of the job: SURVEY 8(d) bytes over the wall clock of the timed region, against N x 8 TB/s
    pipe_bytes = ALG_BYTES_PER_KMER * r["kmers_inserted"] + ALG_BYTES_PER_NOVEL * r["distinct_kmers_total"]
    ach = pipe_bytes / r["seconds"] / 1e9
    out = {"metric": "k-mers/s inserted (build), k=31, 50M x 150bp synthetic reads",
           "value": r["value"], "unit": "k-mers/s", "n_gpus": world, "steps": nsteps, "warmup": nwarm,
           "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": scaling,
           "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": cfg,
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * world), "traffic": None,
                        "what": "SURVEY 8(d): (21.25 B x k-mer occurrences + 8 B x novel keys) / wall clock of the timed region (max over ranks), against %d x 8 TB/s; "
                                "per-stage spans per rank: config.exchange_formats" % world,
                        "alg_bytes": pipe_bytes, "seconds": r["seconds"]},
           "multi_gpu": {"transport": transport, "exchange_format": best, "per_rank": r["per_rank"]},
           "summary": {"value_gkmers_per_s": round(r["value"] / 1e9, 2),
                       **{"%s_gkmers_per_s" % f: round(v["value"] / 1e9, 2) for f, v in one.items()},
                       **({"C5_gkmers_per_s": round(c5["value"] / 1e9, 2)} if c5 else {}),
                       **({"inprocess_multi_gkmers_per_s": round(cfg["inprocess_multi"]["value"] / 1e9, 2)} if "value" in cfg.get("inprocess_multi", {}) else {})}}
    for key in ("checksum_matches_n1", "formats_agree"):
        if key in cfg:
            out["summary"][key] = cfg[key]
    cfg["summary"] = out["summary"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-reads", type=int, default=BATCH_READS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the device-resident figure (profiling passes)")
    ap.add_argument("--oracle-steps", type=int, default=5,
                    help="steps of the workload the CPU oracle builds into the same -n table (timed = cpu_baseline.value; its records' checksum "
                         "is compared with the GPU graph of the same steps).  About 20 s of 32 host threads per step; 10 = the whole 50M-read set")
    ap.add_argument("--no-full-e2e", action="store_true", help="skip e2e_full (the CLI on the whole 50M-read FASTQ: a 15 GB scratch file)")
    ap.add_argument("--input", choices=("packed", "ascii"), default="ascii",
                    help="form of the resident stream: ASCII (default: the 2-bit packing is inside the clock), or the packed "
                         "form the host entry stages (2-bit codes + invalid flags; reported as packed_resident by default)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the same reads and the same total table as N = 1 (config C3); weak = per-GPU work fixed")
    ap.add_argument("--table-slots", type=int, default=TABLE_SLOTS, help="experiments only")
    ap.add_argument("--genome", type=int, default=GENOME_PER_GPU, help="experiments only")
    ap.add_argument("--err", type=float, default=0.001, help="experiments only")
    ap.add_argument("--iid", action="store_true", help="experiments only: C2-stress, iid random reads (every k-mer novel)")
    ap.add_argument("--direct", action="store_true", help="insert with HBM atomics instead of partition + LDS insert")
    ap.add_argument("--defer-tuples", type=int, default=0,
                    help="k-mer occurrences buffered per flush; default: the whole timed region in ONE flush (one pass over the "
                         "table) when that fits 16 G occurrences = 136 GB of the part's 288 GB, never below %d" % DEFER_TUPLES)
    ap.add_argument("--inprocess-multi", type=int, default=0,
                    help="(used by the N > 1 run itself, as a subprocess) the C ABI's one-process table over devices 0..N-1 "
                         "(mcx_graph_create_multi: what `mccortex31 build -D 0,1,..` runs) on the same reads; prints its own JSON line")
    args = ap.parse_args()
    if not args.defer_tuples:
        # One flush for the timed region: every flush streams the 16 GiB table through LDS once, and the part has the HBM
        # for it (bins 8.5 B per buffered occurrence).  A launch is booked with one occurrence per START POSITION of its
        # piece of stream (151 per 120 k-mers here) and the excess comes off the books when the launch has settled --
        # before a full-looking window costs a flush the library waits for that (settle_for_room), so the window is counted
        # in k-mers plus the launches that may be in flight: 20 steps x 5 M reads = 12.0 G -> 12.4 G (105 GB of bins).
        occ = args.steps * args.batch_reads * (READ_LEN - K + 1)
        args.defer_tuples = max(DEFER_TUPLES, min(int(occ * 1.035), 16_000_000_000))
    global HEADLINE_DEFER
    HEADLINE_DEFER = args.defer_tuples

    # stdout carries exactly ONE line, the JSON record: everything else that writes to fd 1 (make,
    # RCCL's version banner, ...) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.inprocess_multi:
        os.write(json_fd, (json.dumps(inprocess_multi(args)) + "\n").encode())
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it: start the N ranks ourselves (one process per GPU,
        # torch.distributed.run on this node) and hand their ONE JSON line (rank 0's) through
        os.dup2(json_fd, 1)
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    from mccortex_amd import shard
    # (MCX_DIST_BACKEND=gloo + MCX_DIST_ONE_DEVICE=0: shard.py's test transport -- several ranks on the one GPU of a
    # test box, collectives staged through host memory; the driver's runs never set them)
    local_rank = shard.local_device(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MCX_BENCH_FORCE_SHARD=1: run the partition -> all-to-all -> insert path even at N=1
    # (validation of the N>1 code on a 1-GPU box; never used for the reported N=1 number)
    force_shard = os.environ.get("MCX_BENCH_FORCE_SHARD") == "1"
    cpu_group = None
    if world > 1 or force_shard:
        os.environ.setdefault("MASTER_PORT", "29531")
        shard.init_process_group(device, rank, world)
        if world > 1:
            import datetime
            cpu_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=30))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import mccortex_amd as mcx

    if world > 1 or force_shard:
        # one process per GPU: both exchange formats, C5, the one-process driver (sharded_main); the rest of this
        # function is the N = 1 run
        out = sharded_main(mcx, shard, dist, args, device, local_rank, rank, world, force_shard)
        if cpu_group is not None:
            dist.barrier(group=cpu_group)   # (a host-side wait: rank 0 may still be timing the one-process driver on these GPUs)
        dist.destroy_process_group()
        if rank == 0:
            sys.stdout.flush()
            sys.stderr.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())  # the ONE JSON line on stdout
        return

    B = args.batch_reads
    nsteps, nwarm = args.steps, args.warmup
    if args.iid:
        batches = [make_batch_iid(B, seed=1000 + i, device=device) for i in range(nsteps + nwarm)]
    else:
        genome = make_genome(args.genome, device, seed=42)
        batches = [make_batch(genome, B, seed=1000 + i, device=device, err_rate=args.err) for i in range(nsteps + nwarm)]
        del genome
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # hand the generator's temporaries back before the graph allocates

    slots_per_gpu = args.table_slots
    # the device's streaming and random-RMW ceilings, measured in this run
    ceil = ceilings(mcx, slots_per_gpu * 16)
    graph = mcx.Graph(K, 1, slots_per_gpu, device=local_rank)
    if args.direct:
        graph.configure("defer", 0)
    else:
        graph.configure("place_bins", PLACE_BINS)
        graph.configure("defer_tuples", args.defer_tuples)
        # the bin workspace is allocated on first use (the library halves the flush size by itself
        # if HBM is short): touch it outside the timed region
        graph.add_stream_dev(0, batches[0][:1024 * (READ_LEN + 1)], 1024 * (READ_LEN + 1))
        graph.sync()
        graph.reset()
    use_packed = args.input == "packed"
    packed = pack_batches(mcx, batches) if use_packed else None
    ext = torch.cuda.ExternalStream(graph.stream, device=device)

    def run_steps(idx):
        for i in idx:
            if use_packed:
                graph.add_packed_dev(0, *packed[i])
            else:
                graph.add_stream_dev(0, batches[i], batches[i].numel())

    def fence():
        torch.cuda.synchronize()
        graph.sync()

    run_steps(range(nsteps, nsteps + nwarm))
    fence()
    graph.reset()
    fence()

    graph.configure("profile", 1)  # HIP events around every kernel launch on the handle's stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(ext)
    run_steps(range(nsteps))
    fence()
    ev1.record(ext)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()

    st = graph.device_stats()
    ist = graph.insert_stats()  # slow-path counters: must be zero on this workload
    kmers_local = st.num_kmers_loaded  # k-mer occurrences this rank k-merised (== inserted job-wide)
    prof = graph.profile()  # {kernel: (launches, total ms)} of the timed region, measured live with HIP events
    cs_local, nodes_local = graph.checksum()   # order-independent checksum of this rank's k-mers
    # Per-kernel durations that mean something: with the flush overlap on (the default, and what `value` is
    # measured with) the insert of one region group runs beside the split of the next, and the HIP-event spans
    # of the two kernels contain each other.  A second pass over the same steps with the overlap off gives
    # isolated durations (one kernel at a time on one stream); roofline.kernels / roofline.dominant use those.
    prof_iso, gpu_ms_iso = None, None
    if not args.direct:
        graph.reset()
        graph.configure("flush_overlap", 0)
        fence()
        graph.configure("profile", 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ext)
        run_steps(range(nsteps))
        fence()
        e1.record(ext)
        torch.cuda.synchronize()
        prof_iso, gpu_ms_iso = graph.profile(), e0.elapsed_time(e1)
        cs_iso, nodes_iso = graph.checksum()
        if (cs_iso, nodes_iso) != (cs_local, nodes_local):
            raise SystemExit("bench: the non-overlapped pass built another graph (%016x / %d against %016x / %d)" % (cs_iso, nodes_iso, cs_local, nodes_local))
        graph.configure("flush_overlap", 1)
    cs_total, nodes_total, kmers_total = cs_local, int(nodes_local), float(kmers_local)

    if rank == 0:
        value = kmers_total / dt
        scaling = "weak"  # (N = 1: the word means nothing here; N > 1 reports what it ran, strong by default)
        shape = "%d reads x %d bp per step from a %d Mbp random genome, table %d slots" % (B, READ_LEN, args.genome // 1_000_000, args.table_slots)
        out = {
            "metric": "k-mers/s inserted (build), k=31, 50M x 150bp synthetic reads",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": nsteps, "warmup": nwarm,
            "ms_per_step": 1e3 * dt / nsteps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("C2-stress: k=31, 1 colour, %d iid random reads x %d bp per step per GPU, table %d slots per GPU; "
                                    "input resident in HBM as an ASCII byte stream" % (B, READ_LEN, args.table_slots)) if args.iid else
                                   "C2: k=31, 1 colour, %s; input already resident in HBM as %s (parse and H2D "
                                   "outside `value`: see host_fed and e2e)" % (shape,
                                                                                "the packed stream mcx_graph_add_reads stages (2-bit codes + invalid flags, 3 bits per position: packing OUTSIDE the clock)"
                                                                                if use_packed else "an ASCII byte stream (1 byte per position: the 2-bit packing is inside the clock)"),
                       "input": "device-resident packed stream (3 bits per position)" if use_packed else "device-resident ASCII stream",
                       "kmer_size": K, "colours": 1, "reads_per_step_per_gpu": B, "read_len": READ_LEN,
                       "table_slots_per_gpu": slots_per_gpu, "table_slots_total": slots_per_gpu,
                       "sharding": "none",
                       "insert_path": "direct HBM atomics" if args.direct else "partition + LDS insert, %d occurrences per flush" % args.defer_tuples,
                       "sub_table_bins": "each half of the flush overlap the best of up to %d allocations by the split's write-pattern probe (mcx_graph_configure place_bins: where the bins lie in HBM decides 13 %% of the split)" % PLACE_BINS,
                       "kmers_inserted": int(kmers_total), "distinct_kmers_rank0": int(st.num_kmers_novel),
                       "distinct_kmers_total": nodes_total, "graph_checksum": "%016x" % cs_total,
                       "table_passes_rank0": ist["flushes"], "fallback_inserts_rank0": ist["fallback_inserts"], "foreign_inserts_rank0": ist["foreign_inserts"]},
        }
        # the figures a reader looks for first, ahead of the long objects (a truncated log still shows them); filled in below
        out["summary"] = {}
        # the same reads must give the same graph whatever N is (strong scaling and N = 1): the sum of the
        # ranks' order-independent checksums against what N = 1 runs of this file report for `steps` steps
        if not args.iid and B == BATCH_READS and args.genome == GENOME_PER_GPU and args.err == 0.001 and READ_LEN == 150 and nsteps in N1_CHECKSUMS:
            out["config"]["checksum_matches_n1"] = ("%016x" % cs_total) == N1_CHECKSUMS[nsteps]
        gpu_ms = ev0.elapsed_time(ev1)
        kprof = prof_iso if prof_iso is not None else prof  # isolated durations where they exist
        dom = max(kprof, key=lambda n: kprof[n][1])
        calls, tot_ms = kprof[dom]
        avg_ms = tot_ms / calls
        # ROOFLINE, as SURVEY.md 8(d) defines it: algorithmic bytes of the path -- 21.25 B per k-mer occurrence
        # (1.25 input + 8 key read + 8 coverage RMW + ~4 edge RMW) + 8 B per novel key -- over the GPU time of
        # the whole timed region (HIP events on the handle's stream), against 8 TB/s.  The path is three kernels
        # of comparable length, so the figure is taken over all of them; each kernel's own design bytes over its
        # own time are under `kernels` (dominant kernel: `dominant`).
        in_b = (0.375 if use_packed else 1.0) * (READ_LEN + 1) / (READ_LEN - K + 1.0)  # stream bytes per occurrence
        kalg = dict(KERNEL_ALG_BYTES, k_stream_bin=in_b + 8.0)
        alg_bytes = kalg[dom] * kmers_local / calls
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        pipe_bytes = ALG_BYTES_PER_KMER * kmers_local + ALG_BYTES_PER_NOVEL * st.num_kmers_novel
        pipe_ach = pipe_bytes / (gpu_ms * 1e-3) / 1e9
        ktab = kernel_table(kprof, kmers_local, 1, in_b)
        # HBM bytes of the whole path per occurrence from the newest PMC summary taken on these kernel sources
        tr_total, tr_src = 0.0, None
        for kn in ktab:
            t_k, src_k = pmc_traffic(kn, kmers_local)
            if t_k is None:
                tr_total, tr_src = None, src_k
                break
            tr_total += t_k
            tr_src = src_k
        out["roofline"] = {"bound": "hbm", "achieved": pipe_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pipe_ach / HBM_PEAK_GBS,
                           "traffic": tr_total, "traffic_source": tr_src,
                           "what": "SURVEY 8(d): (21.25 B x k-mer occurrences + 8 B x novel keys) / GPU time of the timed region; "
                                   "traffic = PMC HBM bytes of the three kernels over the same region (null: no summary for these sources)",
                           "alg_bytes": pipe_bytes, "gpu_ms": gpu_ms, "alg_bytes_per_kmer": ALG_BYTES_PER_KMER, "alg_bytes_per_novel_key": ALG_BYTES_PER_NOVEL,
                           "kernels": ktab,
                           "kernels_what": ("ISOLATED durations: a second pass over the same steps with the flush overlap off (one kernel at a time on one stream, "
                                            "HIP events around every launch; same graph checksum); their sum is the GPU time of that pass (gpu_ms_isolated_pass), "
                                            "the timed region itself (gpu_ms) runs with the overlap on") if prof_iso is not None else "HIP-event spans of the timed region",
                           "gpu_ms_isolated_pass": gpu_ms_iso,
                           "kernels_sum_ms": round(sum(v["total_ms"] for v in ktab.values()), 3),
                           "spans_overlapped": {n: {"launches": c, "total_ms": round(t, 3)} for n, (c, t) in prof.items()} if prof_iso is not None else None,
                           "dominant": {"kernel": dom, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_kernel_ms": avg_ms, "launches": calls,
                                        "alg_bytes_per_launch": alg_bytes,
                                        "note": "the kernel with the largest isolated total; its OWN design bytes (stream + packed tuples) over its average isolated launch"}}
        out["roofline"].update(ceil)
        if ceil.get("measured_copy_gbs"):
            out["roofline"]["frac_of_measured_copy"] = pipe_ach / ceil["measured_copy_gbs"]
        # SURVEY 8(d)'s second ceiling: one random 64-byte sector RMW per occurrence is what the
        # reference's algorithm (and the direct path here) costs; the chip does 17.3 G of those per
        # second on a 16 GiB table (tools/ubench_atomics*.hip, profiles/r01_ubench_atomics*.log).
        # The partition + LDS-insert path is not bound by it: that is the point of the design.
        rmw_peak = ceil.get("measured_random_rmw_per_s") or 17.3e9
        out["roofline"]["random_access"] = {"occurrences_per_s": kmers_local / (gpu_ms * 1e-3),
                                            "measured_random_rmw_peak_per_s": rmw_peak,
                                            "peak_source": "mcx_ubench_random_rmw in this run" if ceil.get("measured_random_rmw_per_s") else "profiles/r01_ubench_atomics.log (round 1)",
                                            "ratio": kmers_local / (gpu_ms * 1e-3) / rmw_peak}
        ex = {}
        if not args.no_extras and not args.iid and not args.direct:
            graph.close()
            torch.cuda.empty_cache()
            osteps = 0 if args.no_cpu_baseline else max(0, min(args.oracle_steps, nsteps))
            ex = extras(mcx, batches, packed, nsteps, args.table_slots, osteps, not args.no_full_e2e)
            for key in ("host_fed", "e2e", "e2e_full", "default_defer", "ascii_resident", "packed_resident", "other_configs", "inprocess_2_shards_1gpu", "inprocess_8_shards_1gpu"):
                if key in ex:
                    out[key] = ex[key]
        if not args.no_cpu_baseline:
            gref = ex.get("_gpu_oracle_steps")
            out["cpu_baseline"] = cpu_baseline(batches[0], ex.get("_fastq_sample"), batches, gref["steps"] if gref else 0, args.table_slots)
            full = out["cpu_baseline"].get("full_size", {})
            if gref and "checksum" in full:
                # the oracle and the GPU built the same steps into the same table size: same records?
                same = full["checksum"] == gref["graph_checksum"] and full["nodes"] == gref["nodes"] and full["kmers"] == gref["kmers"]
                out["config"]["checksum_matches_oracle"] = bool(same)
                out["config"]["oracle_check"] = {"steps": gref["steps"], "of_steps": nsteps, "gpu_checksum": gref["graph_checksum"], "oracle_checksum": full["checksum"],
                                                 "gpu_nodes": gref["nodes"], "oracle_nodes": full["nodes"], "kmers": gref["kmers"],
                                                 "covers_the_headline_graph": gref["steps"] == nsteps and gref["graph_checksum"] == out["config"]["graph_checksum"]}
        if ex.get("_tmpdir"):
            import shutil
            shutil.rmtree(ex["_tmpdir"], ignore_errors=True)
        sm = out["summary"]
        sm["value_gkmers_per_s"] = round(value / 1e9, 2)
        sm["roofline_frac"] = round(out["roofline"]["frac"], 4)
        for key in ("checksum_matches_oracle", "checksum_matches_n1"):
            if key in out["config"]:
                sm[key] = out["config"][key]
        def _g(d, *path):
            for p_ in path:
                d = d.get(p_) if isinstance(d, dict) else None
            return d
        for name, path, scale in (("host_fed_gkmers_per_s", ("host_fed", "value"), 1e-9), ("e2e_seconds_10M_reads", ("e2e", "seconds"), 1),
                                  ("e2e_full_seconds_50M_reads", ("e2e_full", "seconds"), 1), ("default_defer_gkmers_per_s", ("default_defer", "value"), 1e-9),
                                  ("packed_resident_gkmers_per_s", ("packed_resident", "value"), 1e-9), ("ascii_resident_gkmers_per_s", ("ascii_resident", "value"), 1e-9),
                                  ("C4_k63_gkmers_per_s", ("other_configs", "C4_k63", "value"), 1e-9), ("C4_k63_roofline_frac", ("other_configs", "C4_k63", "roofline_frac"), 1),
                                  ("C5_like_gkmers_per_s", ("other_configs", "C5_like_4_colours_1gpu", "value"), 1e-9),
                                  ("C5_interleaved_gkmers_per_s", ("other_configs", "C5_like_interleaved_colours_1gpu", "value"), 1e-9),
                                  ("C2_stress_gkmers_per_s", ("other_configs", "C2_stress", "value"), 1e-9), ("hashtest_ginserts_per_s", ("other_configs", "hashtest", "value"), 1e-9),
                                  ("inprocess_2_shards_1gpu_gkmers_per_s", ("inprocess_2_shards_1gpu", "value"), 1e-9),
                                  ("inprocess_8_shards_1gpu_gkmers_per_s", ("inprocess_8_shards_1gpu", "value"), 1e-9),
                                  ("cpu_baseline_mkmers_per_s_port", ("cpu_baseline", "value"), 1e-6)):
            v = _g(out, *path)
            if isinstance(v, (int, float)):
                sm[name] = round(v * scale, 3)
        v = _g(out, "e2e_full", "checksum_matches_device_resident_build")
        if v is not None:
            sm["e2e_full_checksum_matches_device_resident_build"] = v
        if "cpu_baseline" in out:
            sm["gpu_over_cpu_port"] = round(value / max(1.0, out["cpu_baseline"]["value"]), 1)
            sm["cpu_threads"], sm["cpu_effective_cores"] = out["cpu_baseline"].get("threads"), out["cpu_baseline"].get("effective_cores")
        # SURVEY 8(d)'s clock ("first batch submitted -> table drained", input in HOST memory) next to the device-resident headline
        hf = _g(out, "host_fed", "value")
        if isinstance(hf, (int, float)):
            novel_per_kmer = st.num_kmers_novel / max(1.0, float(kmers_total))
            sm["host_fed_roofline_frac"] = round(hf * (ALG_BYTES_PER_KMER + ALG_BYTES_PER_NOVEL * novel_per_kmer) / 1e9 / HBM_PEAK_GBS, 4)
        for name, path, scale in (("C2_stress_roofline_frac", ("other_configs", "C2_stress", "roofline_frac"), 1), ("hashtest_roofline_frac", ("other_configs", "hashtest", "roofline_frac"), 1),
                                  ("C5_like_roofline_frac", ("other_configs", "C5_like_4_colours_1gpu", "roofline_frac"), 1),
                                  ("C5_reference_gkmers_per_s", ("other_configs", "C5_reference_1gpu", "value"), 1e-9)):
            v = _g(out, *path)
            if isinstance(v, (int, float)):
                sm[name] = round(v * scale, 4)
        v = _g(out, "other_configs", "C5_reference_1gpu", "graph_checksum")
        if v:
            sm["C5_reference_checksum"] = v
        out["config"]["summary"] = sm   # (the driver's record keeps `config`, `roofline`, `cpu_baseline` whole and only the NAMES of other keys)
        out["roofline"]["host_fed"] = {"kmers_per_s": hf, "frac": sm.get("host_fed_roofline_frac"),
                                       "what": "the SURVEY 8(d) clock proper: reads in host memory -> first batch submitted -> table drained (staging threads, PCIe and kernels inside)"}
    if rank == 0:
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())  # the ONE JSON line on stdout


if __name__ == "__main__":
    main()
