#!/bin/bash
# round 4, sixteenth GPU session: the final code -- host-entry / CLI / ABI tests with three staging pairs, rocprofv3 stats
# + PMC passes (profiles/r04q_*), the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r04q; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_hostfed.py tests/test_cli.py tests/test_abi.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
bash tools/prof.sh r04q > $O/prof.log 2>&1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -4 $O/pytest.log; tail -12 $O/prof.log; cat $O/bench.time; head -c 400 $O/bench.json
