#!/bin/bash
# round 4, sixth GPU session: host entry with the copies on their own stream and the background flush driven by the
# measured wait of the compute stream; PCIe rate of the box
cd "$(dirname "$0")/.."
O=gpurun_out/r04f; mkdir -p $O
timeout 300 python tools/ubench_h2d.py > $O/h2d.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_hostfed.py tests/test_cli.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 1500 python tools/exp_hostfed10.py - - MCX_IDLE_FLUSH=0 MCX_STAGE_THREADS=12 MCX_STAGE_THREADS=16 > $O/hostfed.log 2>&1
cat $O/h2d.log; tail -4 $O/pytest.log; cat $O/hostfed.log
