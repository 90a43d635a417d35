#!/bin/bash
# The whole CLI build on a synthetic FASTQ: four timed runs in a row (MCX_TIMING=1), or with PROF=1 one run under rocprofv3
# (kernel, copy and HIP-API statistics + the device busy time over the build: profiles/r05i_cli_*).  Round 5.
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}/prof_cli; mkdir -p $T gpurun_out/prof_cli
(cd $T && python /root/repo/tools/exp_parse_gen.py ${1:-40000000})
export MCX_TIMING=1
mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "timing" | grep -v "export\|epoch" | tr '\n' ' '; echo
rm -f $T/out.ctx; sleep 2
cd /tmp && export TMPDIR=/tmp
if [ "${PROF:-0}" = "1" ]; then
  # (MCX_KEEP_DESTROY=1: the command leaves through _exit() otherwise and rocprofv3 never gets to write its tables)
  MCX_KEEP_DESTROY=1 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_cli -o cli -- /root/repo/mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx > $T/rp.log 2>&1; grep -v "^\[2\|timing\]   export" $T/rp.log | tail -20 | cut -c1-300
else
  for i in 1 2 3; do rm -f $T/out.ctx; sleep 2; /root/repo/mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "timing" | grep -v "export  \|arguments\|insert paths\|add_reads" | tr "\n" " " | sed "s/\[timing\]//g"; echo; done
fi
cd /root/repo
find gpurun_out/prof_cli -name "*stats*" | head
for f in $(find gpurun_out/prof_cli -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-220; done
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_cli/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("kernels: %d launches, first start -> last end %.1f ms, union of kernel intervals %.1f ms, sum of durations %.1f ms" % (
        len(rows), (iv[-1][1] - iv[0][0]) / 1e6, busy / 1e6, sum(e - s for s, e in iv) / 1e6))
    t0 = iv[0][0]
    import collections
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")[:40]
        a = by.setdefault(n, [0, int(r["Start_Timestamp"]), 0, 0])
        a[0] += 1; a[2] = int(r["End_Timestamp"]); a[3] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, s0, e1, tot) in by.items():
        print("  %-42s %5d launches  first start %8.1f ms  last end %8.1f ms  busy %7.1f ms" % (n, c, (s0 - t0) / 1e6, (e1 - t0) / 1e6, tot / 1e6))
for f in glob.glob("gpurun_out/prof_cli/**/*hip_api_stats.csv", recursive=True):
    print("== hip api (top 14 by total)")
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 14: print("  %-40s calls %6s total %9.1f ms  avg %9.1f us" % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find gpurun_out/prof_cli -name "*trace.csv" -delete; find gpurun_out/prof_cli -name "*agent_info.csv" -delete
rm -rf $T
