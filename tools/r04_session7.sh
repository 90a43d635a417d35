#!/bin/bash
# round 4, seventh GPU session: where the host entry's time goes (CPU seconds, cgroup throttling) per packer / thread count
cd "$(dirname "$0")/.."
O=gpurun_out/r04g; mkdir -p $O
cat /sys/fs/cgroup/cpu.stat > $O/cpustat0.log 2>&1
timeout 2000 python tools/exp_hostfed10.py - MCX_FUSED_PACK=0 MCX_STAGE_THREADS=8 MCX_STAGE_THREADS=12 MCX_STAGE_THREADS=15 MCX_STAGE_THREADS=16 MCX_STAGE_THREADS=32 MCX_FUSED_PACK=0,MCX_STAGE_THREADS=15 - > $O/hostfed.log 2>&1
cat $O/hostfed.log
