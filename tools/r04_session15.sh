#!/bin/bash
# round 4, fifteenth GPU session: C4 (k = 63) under region / sub-table balance and overlap; host entry with three staging pairs
cd "$(dirname "$0")/.."
O=gpurun_out/r04p; mkdir -p $O
for v in "" "MCX_FLUSH_OVERLAP=0" "MCX_LB1=10" "MCX_LB1=10 MCX_FLUSH_OVERLAP=0" "MCX_LB1=8" "MCX_FLUSH_REGIONS=16" "MCX_FLUSH_REGIONS=64"; do
  echo "== $v" >> $O/c4.log
  env $v timeout 600 python tools/exp_c4c5.py c4 >> $O/c4.log 2>&1
done
export MCX_STAGE_TIMING=1 REPS=3
timeout 1500 python tools/exp_hostfed10.py - - MCX_STAGE_THREADS=12 MCX_STAGE_THREADS=16 > $O/hostfed.log 2>&1
grep -v amdgpu.ids $O/c4.log; cat $O/hostfed.log
