#!/bin/bash
# Run bench.py once per tuning variant in build/variants (on the GPU box); one line per variant with
# the whole-job rate and every kernel's total time in the timed region.
# Usage: [STEPS=10] [ONLY="a b"] tools/sweep.sh [extra bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
shopt -s nullglob
for so in "" build/variants/lib_*.so; do
  name=${so:-default}
  if [ -n "$ONLY" ] && [ -n "$so" ]; then
    keep=0; for o in $ONLY; do [ "$so" = "build/variants/lib_$o.so" ] && keep=1; done
    [ $keep = 1 ] || continue
  fi
  MCX_LIB=${so:+$PWD/$so} timeout 600 python bench.py --steps ${STEPS:-10} --warmup 1 --no-cpu-baseline --no-extras "$@" 2>gpurun_out/sweep_last.err | tail -1 | \
    python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read())
    ks=d['roofline']['kernels']
    print('%-28s %6.2f G/s  %6.3f ms/step  pipe %.3f  ' % ('$name', d['value']/1e9, d['ms_per_step'], d['roofline']['frac']) + '  '.join('%s %.2f' % (k.replace('k_',''), v['total_ms']) for k,v in ks.items()) + '  cs ' + d['config']['graph_checksum'])
except Exception as e:
    print('%-28s FAILED %s' % ('$name', e))
"
done | tee -a gpurun_out/sweep.log
