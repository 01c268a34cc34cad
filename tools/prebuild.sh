#!/bin/bash
# Builds the production library and fails loudly (exit 1) when any source does not compile: run before every gpurun call
# (the GPU box would otherwise spend its minutes re-discovering the compile error).
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from propainter_amd import build
import ctypes
path = build.build()
ctypes.CDLL(path)
print("built", path)
PY
