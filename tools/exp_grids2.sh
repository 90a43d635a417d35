#!/bin/bash
# round 6: launch grids of the split and the insert (blocks), C2 bench, 10 steps
run() { env "$@" python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$*', json.dumps({'ms_per_step': round(d['ms_per_step'], 3), 'kernels': {k: v['total_ms'] for k, v in d['roofline'].get('kernels', {}).items()}}))
"; }
run MCX_X=0
for gi in 2048 4096 16384; do run MCX_GRID_INSERT=$gi; done
for gs in 1024 4096 8192; do run MCX_GRID_SPLIT=$gs; done
for gk in 1024 4096; do run MCX_GRID_STREAM=$gk; done
