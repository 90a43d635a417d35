#!/bin/bash
# round 4, twentieth GPU session: lds_apply with a whole bucket per step: parity (high load factors, hot keys, fuzz), then
# hashtest / C2-stress / C2 / C4
cd "$(dirname "$0")/.."
O=gpurun_out/r04u; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_ctxio.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
ONLY=none STEPS=10 tools/sweep.sh > $O/sweep.log 2>&1
ONLY=none STEPS=10 tools/sweep.sh >> $O/sweep.log 2>&1
ONLY=none STEPS=10 tools/sweep.sh --iid --table-slots 8589934592 --defer-tuples 7620000000 >> $O/sweep.log 2>&1
C4_DEFER=7700000000 MCX_FLUSH_OVERLAP=0 python tools/exp_c4c5.py c4 2>&1 | grep C4 >> $O/sweep.log
python - >> $O/sweep.log 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench, mccortex_amd as mcx
r = bench.hashtest(mcx, torch.device("cuda", 0), 1 << 30)
print("hashtest %.2f G inserts/s, %.1f ms, %s, all_inserted_once %s" % (r["value"] / 1e9, r["seconds"] * 1e3, r["kernels"], r["all_inserted_once"]))
PY
tail -4 $O/pytest.log; cat $O/sweep.log
