#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes for bench.py.
# Usage: bash tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { # name, rocprof flags...
  local name=$1; shift
  rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- python $REPO/bench.py $ARGS > $OUT/$name.log 2>&1
  echo "== $name rc=$?"; tail -2 $OUT/$name.log | cut -c1-400
}
run trace --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_ea --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run pmc_atomic --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_READ_sum
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU
rocprofv3 -L > $OUT/counters_list.txt 2>&1
find $OUT -name "*.csv" | head -30
# keep only small summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
