#!/bin/bash
# round 4, second GPU session: per-phase times of the build kernels, the host entry with the AVX-512 packer,
# the bench with the oracle on the WHOLE 50 M-read set
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
MCX_LIB=$PWD/build/variants/lib_phases.so timeout 600 python tools/exp_phases.py > $O/phases.log 2> $O/phases.err
timeout 900 python tools/exp_hostfed10.py - MCX_NO_AVX512=1 MCX_STAGE_THREADS=16 MCX_STAGE_THREADS=32 MCX_STAGE_THREADS=48 > $O/hostfed.log 2>&1
( time timeout 1200 python bench.py --steps 10 --warmup 1 --oracle-steps 10 > $O/bench_oracle10.json 2> $O/bench.err ) 2> $O/bench.time
cat $O/phases.log; cat $O/hostfed.log; cat $O/bench.time
