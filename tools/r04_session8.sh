#!/bin/bash
# round 4, eighth GPU session: per-phase wall clock of the host entry (MCX_STAGE_TIMING)
cd "$(dirname "$0")/.."
O=gpurun_out/r04h; mkdir -p $O
export MCX_STAGE_TIMING=1 REPS=3
timeout 2000 python tools/exp_hostfed10.py - MCX_FUSED_PACK=0 MCX_STAGE_THREADS=16 MCX_FUSED_PACK=0,MCX_STAGE_THREADS=16 MCX_IDLE_FLUSH=0 MCX_STAGE_BYTES=33554432 MCX_STAGE_BYTES=268435456 > $O/hostfed.log 2>&1
cat $O/hostfed.log
