#!/bin/bash
# Split geometry: 512 threads x 16 tuples (tiles of 8192, the default for one-word tuples) against 256 x 16.
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-12s' % '$1', round(d['value']/1e9,1), round(d['ms_per_step'],3), {k:round(v['total_ms'],1) for k,v in d['roofline']['kernels'].items()}, d['config']['graph_checksum'])"; }
for i in 1 2; do
MCX_SPLIT_T=512 run wide
MCX_SPLIT_T=256 run narrow
done
