#!/bin/bash
# Builds a VARIANT of the library for in-call A/B runs: tools/build_variant.sh <name> [extra hipcc flags...] -> build/<name>/libpropainter_hip.so
# (PP_DIAG kernels included so tools/kbench can drive it: LD_LIBRARY_PATH=build/<name> build/kbench ...).  Tuning tool, not product.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$R/build/$NAME
mkdir -p $OUT/obj
pids=()
for s in $R/propainter_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DPP_DIAG "$@" -c $s -o $OUT/obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libpropainter_hip.so $OUT/obj/*.o
echo built $OUT/libpropainter_hip.so
