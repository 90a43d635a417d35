#!/bin/bash
# Run bench.py once per tuning variant in build/variants (on the GPU box).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for so in "" build/variants/lib_*.so; do
  name=${so:-default}
  MCX_LIB=${so:+$PWD/$so} python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.3e kmers/s  kernel %.2f ms' % ('$name', d['value'], d['roofline']['avg_kernel_ms']))"
done | tee gpurun_out/sweep.log
