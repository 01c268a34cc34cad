#!/bin/bash
# parity suite + headline bench (no profiles)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -12
bash tools/gpu_bench.sh --steps 3 --warmup 1 --no-cpu-baseline "$@"
