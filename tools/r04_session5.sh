#!/bin/bash
# round 4, fifth GPU session: the one-pass host packer (reads -> packed chunk, no intermediate ASCII block):
# parity of the host entry, then host_fed under packer / thread-count variants
cd "$(dirname "$0")/.."
O=gpurun_out/r04e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_hostfed.py tests/test_abi.py tests/test_cli.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 1500 python tools/exp_hostfed10.py - MCX_FUSED_PACK=0 MCX_STAGE_THREADS=8 MCX_STAGE_THREADS=12 MCX_STAGE_THREADS=16 MCX_STAGE_THREADS=32 MCX_FUSED_PACK=0,MCX_STAGE_THREADS=16 > $O/hostfed.log 2>&1
tail -4 $O/pytest.log; cat $O/hostfed.log
