#!/bin/bash
# Launch-geometry sweep of the three build kernels on the bench shape (environment knobs of the library).
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-40s' % '$1', round(d['value']/1e9,1), round(d['ms_per_step'],3), {k:round(v['total_ms'],1) for k,v in d['roofline']['kernels'].items()})"; }
run base
MCX_GRID_STREAM=1024 run stream1024
MCX_GRID_STREAM=4096 run stream4096
MCX_GRID_SPLIT=1024 run split1024
MCX_GRID_SPLIT=4096 run split4096
MCX_GRID_INSERT=512 run insert512
MCX_GRID_INSERT=2048 run insert2048
MCX_FLUSH_REGIONS=16 run regions16
MCX_FLUSH_REGIONS=64 run regions64
MCX_FLUSH_REGIONS=128 run regions128
run base_again
