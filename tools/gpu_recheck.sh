#!/bin/bash
# Quick re-check of the shipped library after a source change that only touched diagnostic code paths: operator / split-plane / C-ABI parity
# tests, smoke, a short headline bench.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_split_plane_gpu.py tests/test_cabi_ref_layout_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/recheck_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/recheck_pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/recheck_smoke.log 2>&1; echo "smoke exit $?"; grep smoke gpurun_out/recheck_smoke.log | tail -3
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precisions --no-profile > gpurun_out/recheck_bench.json 2> gpurun_out/recheck_bench.err
echo "bench exit $?"; python -c "import json; d=json.load(open('gpurun_out/recheck_bench.json')); print(d['value'], d['ms_per_step'])"
