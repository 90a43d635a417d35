#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_*.csv|md."""
import collections
import csv
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(n):
    return n if len(n) < 90 else n[:87] + "..."


lines = ["# rocprofv3 summary `%s`" % tag, "",
         "Command: `python bench.py %s` under `tools/prof.sh` (one rocprofv3 run per pass)." %
         (" ".join(sys.argv[2:]) or "--steps 3 --warmup 1 --no-cpu-baseline"), ""]
ks = os.path.join(src, "trace", "trace_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    lines += ["## `--kernel-trace --stats` (top kernels by total time)", "",
              "| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:8]:
        lines.append("| `%s` | %s | %.3f | %.3f | %.3f | %.3f | %s |" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6,
            float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(rows[0].keys())
        for r in rows[:12]:
            r = dict(r); r["Name"] = short(r["Name"]); w.writerow(r.values())
lines += ["", "## PMC passes (per launch of the build kernels, last four launches of the run; separate rocprofv3 runs)", "",
          "| pass | counter | per-launch values |", "|---|---|---|"]
for d in sorted(os.listdir(src)):
    cc = os.path.join(src, d, d + "_counter_collection.csv")
    if not os.path.exists(cc):
        continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        kn = r["Kernel_Name"]
        if "mcx::k_" in kn:
            key = (kn.split("(")[0].replace("void ", ""), r["Counter_Name"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
    for (kn, c), v in agg.items():
        v = [x for x in v if x > 0.01 * max(v)] or v  # drop the tiny warm-up launches
        lines.append("| %s | `%s` %s | %s |" % (d, short(kn), c, ", ".join("%.4g" % x for x in v[-4:])))  # the last launches: the timed region's
# HBM traffic of the build kernels from the FETCH_SIZE / WRITE_SIZE passes, corrected as
# MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports half of
# the bytes of wide coalesced reads (x2).  Cross-check in DESIGN.md section 5: with the x2 the reads
# of all three kernels equal their algorithmic reads, and WRITE_SIZE of the LDS insert equals the
# table size (written back exactly once).
import json
# launches of the timed region: from the bench record printed in the same pass
def bench_record(passname):
    log = os.path.join(src, passname + ".log")
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith('{"metric"'):
                return json.loads(line)
    return None

traffic = {}
for passname, counter, scale in (("pmc_fetch", "FETCH_SIZE", 2 * 1024.0), ("pmc_write", "WRITE_SIZE", 1024.0)):
    cc = os.path.join(src, passname, passname + "_counter_collection.csv")
    rec = bench_record(passname)
    if not os.path.exists(cc) or rec is None:
        continue
    timed = {k: v["launches"] for k, v in rec["roofline"]["kernels"].items()}
    per = collections.OrderedDict()
    rows = [r for r in csv.DictReader(open(cc)) if "mcx::k_" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0))
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcx::", "").split("<")[0]
        per.setdefault(name, []).append(float(r["Counter_Value"]) * scale)
    for name, vals in per.items():
        n = timed.get(name)
        if not n:
            continue
        t = traffic.setdefault(name, {"launches": n, "occurrences": rec["config"]["kmers_inserted"]})
        t["read_bytes" if counter == "FETCH_SIZE" else "written_bytes"] = sum(vals[-n:])  # the timed region's launches
if traffic:
    sys.path.insert(0, os.getcwd())
    import bench
    meta = {"command": "python bench.py " + " ".join(sys.argv[2:]),
            "csrc_digest": bench.csrc_digest(),  # bench.py refuses the figure once the kernel sources change
            "note": "launches of the timed region only; read = 2 * FETCH_SIZE * 1024, written = WRITE_SIZE * 1024 (separate --pmc passes)",
            "kernels": traffic}
    json.dump(meta, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
    lines += ["", "## HBM traffic of the timed region (FETCH_SIZE x 2 and WRITE_SIZE, KiB -> bytes)", "",
              "| kernel | launches | read GB | written GB | bytes per occurrence |", "|---|---|---|---|---|"]
    for k, t in traffic.items():
        rd, wr = t.get("read_bytes", 0.0), t.get("written_bytes", 0.0)
        lines.append("| `%s` | %d | %.2f | %.2f | %.2f |" % (k, t["launches"], rd / 1e9, wr / 1e9, (rd + wr) / t["occurrences"]))
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
