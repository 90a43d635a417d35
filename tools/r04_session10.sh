#!/bin/bash
# round 4, tenth GPU session: does NUMA placement bound the host entry?  whole process pinned to node 0 / node 1
cd "$(dirname "$0")/.."
O=gpurun_out/r04j; mkdir -p $O
export MCX_STAGE_TIMING=1 REPS=3
timeout 2000 python tools/exp_hostfed10.py - PIN_NODE=0 PIN_NODE=1 PIN_NODE=0,MCX_STAGE_THREADS=12 PIN_NODE=0,MCX_STAGE_THREADS=16 PIN_NODE=0,MCX_IDLE_FLUSH=0 - > $O/hostfed.log 2>&1
cat $O/hostfed.log
