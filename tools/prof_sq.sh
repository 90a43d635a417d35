#!/bin/bash
# usage: prof_sq.sh <tag>   (forced-shard bench, SQ + LDS counters of the owner-side kernel)
TAG=$1
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp MCX_BENCH_FORCE_SHARD=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for pass in "pmc_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU" "pmc_lds SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  python - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/$name/${name}_counter_collection.csv")):
    if "k_superk_bin" in r["Kernel_Name"] or "k_stream_superk" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    v = [x for x in v if x > 0.01 * max(v)] or v
    print("$TAG", k, c, "%.4g" % (sum(v) / len(v)), len(v))
PY
  find $OUT -name "*.csv" -size +1M -delete
done
