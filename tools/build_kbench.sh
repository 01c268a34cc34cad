#!/bin/bash
# Builds build/kbench (stand-alone conv micro-benchmark) against the in-tree libpropainter_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
PP_DIAG=1 python -c "import sys; sys.path.insert(0, '$R'); from propainter_amd import build; print(build.build())"
mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $R/tools/kbench.cpp -o $R/build/kbench -L$R/propainter_amd/lib -lpropainter_hip -Wl,-rpath,'$ORIGIN/../propainter_amd/lib'
echo built $R/build/kbench
