#!/bin/bash
# round 4, thirteenth GPU session: C2-stress (iid reads, 2^33 slots): where the split and the insert spend their time; 2048 regions
cd "$(dirname "$0")/.."
O=gpurun_out/r04n; mkdir -p $O
PHASES_CFG=stress MCX_LIB=$PWD/build/variants/lib_phases.so timeout 900 python tools/exp_phases.py > $O/phases_stress.log 2> $O/phases.err
ONLY=none STEPS=10 tools/sweep.sh --iid --table-slots 8589934592 --defer-tuples 6300000000 > $O/sweep.log 2>&1
MCX_LB1=11 ONLY=none STEPS=10 tools/sweep.sh --iid --table-slots 8589934592 --defer-tuples 6300000000 >> $O/sweep.log 2>&1
cat $O/phases_stress.log; tail -3 $O/phases.err; cat $O/sweep.log
