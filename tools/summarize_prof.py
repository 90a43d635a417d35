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
lines += ["", "## PMC passes (per launch of the build kernels; separate rocprofv3 runs)", "",
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
        lines.append("| %s | `%s` %s | %s |" % (d, short(kn), c, ", ".join("%.4g" % x for x in v[:4])))
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
