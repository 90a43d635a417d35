#!/bin/bash
# round 4, nineteenth GPU session: soak of the other randomised parity tests (arbitrary bytes and ragged lengths through
# the host entry with the one-pass packer, super-k-mer records on arbitrary bytes, the PCR filter's fuzz), then the
# final profile passes and the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r04v; mkdir -p $O
( time MCX_FUZZ_SEEDS_BYTES=${SEEDS:-20} timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_pcr.py -m gpu -x -q -k "arbitrary_bytes or fuzz_arbitrary" ) > $O/soak_bytes.log 2>&1
echo "soak rc $?" >> $O/soak_bytes.log
bash tools/prof.sh r04v > $O/prof.log 2>&1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -4 $O/soak_bytes.log; tail -3 $O/prof.log; cat $O/bench.time; head -c 300 $O/bench.json
