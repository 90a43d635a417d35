#!/bin/bash
# round 6: k_stream_fc (fixed-capacity LDS segments, mcx_streamfc.h) against k_stream_bin (LDS sort): parity, the C2 bench per
# capacity, phase clocks of both (build/variants/lib_phases.so = tools/variants.sh phases:"-DMCX_PHASES")
mkdir -p gpurun_out
O=gpurun_out/${OUT:-r06c_fc}.log
: > $O
if [ -z "$SKIP_PARITY" ]; then
echo "== parity with MCX_STREAM_FC=10 $ENVX" >> $O
env $ENVX MCX_STREAM_FC=10 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_edges_pin.py -m gpu -x -q 2>&1 | tail -5 >> $O
fi
for cap in ${CAPS:-0 10}; do
  echo "== bench MCX_STREAM_FC=$cap $ENVX" >> $O
  env $ENVX MCX_STREAM_FC=$cap timeout 600 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras 2>>$O.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'ms_per_step': d['ms_per_step'], 'value': d['value'], 'frac': d['roofline']['frac'], 'checksum': d['config']['graph_checksum'], 'fallback': d['config'].get('fallback_inserts_rank0'), 'kernels': d['roofline'].get('kernels')}))
" >> $O
done
if [ -f build/variants/lib_phases.so ]; then
  for cap in ${CAPS:-0 10}; do
    echo "== phases MCX_STREAM_FC=$cap $ENVX" >> $O
    env $ENVX MCX_STREAM_FC=$cap MCX_LIB=$PWD/build/variants/lib_phases.so timeout 600 python tools/exp_phases.py > $O.ph 2>&1; grep -A 9 "^== k = 31" $O.ph | head -12 >> $O; tail -3 $O.ph >> $O.err
  done
fi
tail -60 $O
