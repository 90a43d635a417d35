#!/bin/bash
# Two PMC passes (SQ instruction mix, LDS) of bench.py for one library build; prints the per-launch
# averages of the kernel whose name matches $KERNEL (default k_stream_bin).
# Usage (on the GPU box): [MCX_LIB=path] [KERNEL=k_stream_bin] bash tools/pmc_quick.sh <tag>
TAG=${1:-q}
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
ARGS="--steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU -d $OUT/sq -o sq --output-format csv -- python $REPO/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM -d $OUT/lds -o lds --output-format csv -- python $REPO/bench.py $ARGS > $OUT/lds.log 2>&1
python - $OUT ${KERNEL:-k_stream_bin} <<'PY'
import sys, csv, glob, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + '/*/*counter_collection.csv') + glob.glob(out + '/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in sorted(acc.items()):
        v = v[len(v) // 2:]  # skip warm-up launches
        print('%-24s %14.4g  (%d launches)' % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -delete
