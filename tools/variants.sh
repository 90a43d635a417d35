#!/bin/bash
# Build tuning variants of the kernel library (compile-time knobs) into build/variants/.
# Usage: tools/variants.sh name:"-DFLAG=.. -DFLAG2=.." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $flags mccortex_amd/csrc/mcx_api.hip -o build/variants/lib_$name.so &
done
wait
ls -la build/variants/
