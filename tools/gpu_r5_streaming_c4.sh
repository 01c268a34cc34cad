#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_modules_gpu.py -x -q -m gpu -k "streaming or sharded" 2>&1 | tail -5 > gpurun_out/r5_streaming_tests.txt
cat gpurun_out/r5_streaming_tests.txt
timeout 1200 python tools/bench_streaming.py --steps 2 > gpurun_out/r5_streaming_config4.json 2> gpurun_out/r5_streaming_config4.err
tail -3 gpurun_out/r5_streaming_config4.err | cut -c1-1500
cat gpurun_out/r5_streaming_config4.json | cut -c1-3000
