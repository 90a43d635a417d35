#!/bin/bash
# round 4, fourth GPU session: the split as it was before 5878bdd (that change cost 2.7 ms), region / sub-table balance
# (MCX_LB1=8: 256 regions x 1024 sub-tables), flush-group size with the overlap on, a flush window small enough for a
# group's sub-table bins to stay in the Infinity Cache, host topology
cd "$(dirname "$0")/.."
O=gpurun_out/r04d; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E 'NUMA|Socket|Model name|Thread|Core'; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3; free -g | head -2; } > $O/host.log 2>&1
ONLY=none STEPS=10 tools/sweep.sh > $O/sweep.log 2>&1
for v in "MCX_LB1=8" "MCX_LB1=8 MCX_FLUSH_REGIONS=16" "MCX_FLUSH_REGIONS=16" "MCX_FLUSH_REGIONS=8" "MCX_FLUSH_REGIONS=64"; do
  echo "== $v" >> $O/sweep.log
  env $v ONLY=none STEPS=10 tools/sweep.sh >> $O/sweep.log 2>&1
done
for dt in 200000000 400000000 800000000; do
  echo "== --defer-tuples $dt" >> $O/sweep.log
  ONLY=none STEPS=10 tools/sweep.sh --defer-tuples $dt >> $O/sweep.log 2>&1
  echo "== --defer-tuples $dt MCX_FLUSH_OVERLAP=0" >> $O/sweep.log
  MCX_FLUSH_OVERLAP=0 ONLY=none STEPS=10 tools/sweep.sh --defer-tuples $dt >> $O/sweep.log 2>&1
done
cat $O/host.log $O/sweep.log
