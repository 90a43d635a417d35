#!/bin/bash
# round 4, ninth GPU session: the expand-based one-pass packer in the host entry
cd "$(dirname "$0")/.."
O=gpurun_out/r04i; mkdir -p $O
export MCX_STAGE_TIMING=1 REPS=3
( time timeout 900 python -m pytest tests/test_gpu_hostfed.py tests/test_abi.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 2000 python tools/exp_hostfed10.py - MCX_FUSED_PACK=0 - MCX_FUSED_PACK=0 MCX_STAGE_THREADS=12 MCX_STAGE_THREADS=16 MCX_FUSED_PACK=0,MCX_STAGE_THREADS=16 > $O/hostfed.log 2>&1
tail -3 $O/pytest.log; cat $O/hostfed.log
