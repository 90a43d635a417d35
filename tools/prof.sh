#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes for bench.py.
# Usage: bash tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 10 --warmup 1 --no-cpu-baseline --no-extras}
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
# the same without the flush overlap (MCX_FLUSH_OVERLAP=0): kernel by kernel, nothing co-running
MCX_FLUSH_OVERLAP=0 run trace_noovl --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_ea --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run pmc_atomic --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_READ_sum
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU
run pmc_lds --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# condense on the box (gpurun copies back at most 64 MiB): summary + small CSVs only
cd $REPO && python tools/summarize_prof.py $TAG $ARGS > $OUT/summary.txt 2>&1
cp $OUT/trace_noovl/trace_noovl_kernel_stats.csv profiles/${TAG}_kernel_stats_no_overlap.csv 2>/dev/null
cp profiles/${TAG}_* $OUT/ 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
