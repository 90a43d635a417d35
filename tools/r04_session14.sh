#!/bin/bash
# round 4, fourteenth GPU session: the whole -m gpu suite, smoke(), and the default bench line (as the driver runs them)
cd "$(dirname "$0")/.."
O=gpurun_out/r04o; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
( time timeout 600 python -c "import __graft_entry__ as e; e.smoke()" ) > $O/smoke.log 2>&1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -5 $O/pytest.log; tail -3 $O/smoke.log; cat $O/bench.time; head -c 800 $O/bench.json
