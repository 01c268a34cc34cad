#!/bin/bash
# First-measurement script: bench at 240p and 720p, per-kernel HIP-event breakdown; leaves JSON under gpurun_out/.
mkdir -p gpurun_out
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.txt 2>&1
timeout 600 python bench.py --height 240 --width 432 --steps 2 --warmup 1 > gpurun_out/bench_240.json 2> gpurun_out/bench_240.err
echo "240p exit $?"; tail -c 3000 gpurun_out/bench_240.json; tail -5 gpurun_out/bench_240.err
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_720.json 2> gpurun_out/bench_720.err
echo "720p exit $?"; tail -c 4000 gpurun_out/bench_720.json; tail -5 gpurun_out/bench_720.err
