#!/bin/bash
# round 4, eighteenth GPU session: soak of the randomised parity tests on the round's final code (random k, colours,
# 1-8 shards, both exchange formats with the copy kernel, pool / flush / piece sizes; arbitrary bytes)
cd "$(dirname "$0")/.."
O=gpurun_out/r04s; mkdir -p $O
( time MCX_FUZZ_SEEDS=${SEEDS:-1500} timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_colours" ) > $O/soak.log 2>&1
echo "soak rc $?" >> $O/soak.log
tail -5 $O/soak.log
