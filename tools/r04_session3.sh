#!/bin/bash
# round 4, third GPU session: window fills in LDS (split), register scatter variant, settled launches, dynamic staging
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_hostfed.py tests/test_abi.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
STEPS=10 ONLY="scatter" tools/sweep.sh > $O/sweep.log 2>&1
STEPS=10 ONLY="scatter" tools/sweep.sh >> $O/sweep.log 2>&1
MCX_LIB=$PWD/build/variants/lib_phases.so timeout 600 python tools/exp_phases.py > $O/phases.log 2> $O/phases.err
( time timeout 900 python bench.py --steps 10 --warmup 1 --oracle-steps 1 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -4 $O/pytest.log; cat $O/sweep.log; cat $O/phases.log; cat $O/bench.time
