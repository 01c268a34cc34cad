#!/bin/bash
# Streaming schedule (sharding.StreamingClipGraph): parity test, then BASELINE config 4 on one GPU against the whole-pass hipGraph.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -x -p no:cacheprovider -k "streaming_schedule or sharded_pass_as_hipgraphs" > gpurun_out/r3z_streaming_pytest.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/r3z_streaming_pytest.log
timeout 900 python tools/bench_streaming.py --steps 2 $STREAM_ARGS > gpurun_out/r3z_streaming_c4.json 2> gpurun_out/r3z_streaming_c4.err
echo "bench exit $?"; cat gpurun_out/r3z_streaming_c4.json; tail -5 gpurun_out/r3z_streaming_c4.err
