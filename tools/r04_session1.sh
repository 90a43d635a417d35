#!/bin/bash
# round 4, first GPU session: the whole -m gpu suite, the copy-ceiling variants, the default bench run
cd "$(dirname "$0")/.."
O=gpurun_out/r04a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 300 ./build/ubench_copy2 > $O/ubench_copy2.log 2>&1
( time timeout 900 python bench.py --steps 10 --warmup 1 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -3 $O/pytest.log; tail -5 $O/ubench_copy2.log; cat $O/bench.time; head -c 600 $O/bench.json
