#!/bin/bash
# round 6: environment knobs against the driver's command (20 steps, one table pass, placed bins)
run() { env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$*', round(d['value'] / 1e9, 2), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), {k: v['total_ms'] for k, v in d['roofline'].get('kernels', {}).items()})
"; }
for v in "$@"; do run $v; done
