#!/bin/bash
# round 6: k_stream_bin against the flush window (= the size of its 8 x 512 output segments) and the number of replicas
# usage: bash tools/exp_window.sh            (windows 8 / 12.4 / 16 G at 8 replicas)
#        REPS="4 8 16" WINS="8000000000 12420000000" bash tools/exp_window.sh
for rep in ${REPS:-8}; do
for w in ${WINS:-8000000000 12420000000 16000000000}; do
  MCX_REP1=$rep python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras --defer-tuples $w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('replicas', $rep, 'window', $w, json.dumps({'ms_per_step': d['ms_per_step'], 'frac': d['roofline']['frac'], 'kernels': {k: v['total_ms'] for k, v in d['roofline'].get('kernels', {}).items()}}))
"
done
done
