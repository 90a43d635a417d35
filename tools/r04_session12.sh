#!/bin/bash
# round 4, twelfth GPU session: k_stream_bin with 512-thread blocks (two tiles sorted as one): parity, then C2
cd "$(dirname "$0")/.."
O=gpurun_out/r04m; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
ONLY=none STEPS=10 tools/sweep.sh > $O/sweep.log 2>&1
MCX_STREAM_T=256 ONLY=none STEPS=10 tools/sweep.sh >> $O/sweep.log 2>&1
ONLY=none STEPS=10 tools/sweep.sh >> $O/sweep.log 2>&1
MCX_STREAM_T=256 ONLY=none STEPS=10 tools/sweep.sh >> $O/sweep.log 2>&1
ONLY=none STEPS=10 tools/sweep.sh --input packed >> $O/sweep.log 2>&1
MCX_STREAM_T=256 ONLY=none STEPS=10 tools/sweep.sh --input packed >> $O/sweep.log 2>&1
tail -4 $O/pytest.log; cat $O/sweep.log
