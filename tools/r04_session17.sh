#!/bin/bash
# round 4, seventeenth GPU session: host entry as a two-chunk pipeline (pack n+1 while chunk n is enqueued): parity, rate
cd "$(dirname "$0")/.."
O=gpurun_out/r04r; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_hostfed.py tests/test_cli.py tests/test_abi.py tests/test_gpu_multi.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
export MCX_STAGE_TIMING=1 REPS=3
timeout 1500 python tools/exp_hostfed10.py - - MCX_STAGE_THREADS=16 MCX_STAGE_THREADS=12 - > $O/hostfed.log 2>&1
tail -4 $O/pytest.log; cat $O/hostfed.log
