#!/bin/bash
# round 4, eleventh GPU session: in-process multi-GPU exchange with the device-side copy of the filled parts
# (k_copy_filled): parity, then all shards on the one GPU against whole-block peer copies
cd "$(dirname "$0")/.."
O=gpurun_out/r04l; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_cli.py tests/test_ctxio.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k multi ) > $O/pytest_full.log 2>&1
echo "pytest rc $?" >> $O/pytest_full.log
timeout 900 python tools/exp_inproc.py > $O/inproc.log 2>&1
tail -4 $O/pytest.log; tail -4 $O/pytest_full.log; cat $O/inproc.log
