#!/bin/bash
# Round 3, session 2: tile sweep of the split-plane layers + the new pyramid / epilogue / upsample paths, then their parity tests.
mkdir -p gpurun_out
timeout 600 python tools/bench_split.py --reps 20 > gpurun_out/r3u_split_sweep.txt 2> gpurun_out/r3u_split_sweep.err
echo "sweep exit $?"; tail -60 gpurun_out/r3u_split_sweep.txt; tail -5 gpurun_out/r3u_split_sweep.err
timeout 600 python -m pytest tests/test_split_plane_gpu.py tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "pyramid or batched_gemm or upsample or raft or aux or corr" > gpurun_out/r3u_pytest.log 2>&1
echo "pytest exit $?"; tail -8 gpurun_out/r3u_pytest.log
