#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "clip_graph or instance_norm or raft" 2>&1 | tail -15
[ -n "$NO_BENCH" ] || bash tools/gpu_bench.sh "$@"
