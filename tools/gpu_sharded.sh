#!/bin/bash
# BASELINE configs 4 / 5 on what the box has: config 4 (720p x 320 frames, sub-video chunks of 80) and config 5 (1080p x 160,
# subvideo_length 20) on ONE GPU, plus the 2-rank variant when two GPUs are visible.
mkdir -p gpurun_out
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "gpus: $NG"
timeout 900 python bench.py --sharded --frames 320 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_bench_c4_1gpu.json 2> gpurun_out/r2_bench_c4_1gpu.err; echo "c4 exit $?"
timeout 900 python bench.py --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_bench_c5_1gpu.json 2> gpurun_out/r2_bench_c5_1gpu.err; echo "c5 exit $?"
for f in c4 c5; do python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2_bench_${f}_1gpu.json'))
    print('$f', d['value'], d['ms_per_step'], d['config']['workload'], d['memory'], d['submission'], d['roofline'])
except Exception as e:
    print('$f parse failed', e); print(open('gpurun_out/r2_bench_${f}_1gpu.err').read()[-2000:])
PY
done
if [ "$NG" -ge 2 ]; then
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --sharded --frames 320 --steps 1 --warmup 0 > gpurun_out/r2_bench_c4_2gpu.json 2> gpurun_out/r2_bench_c4_2gpu.err; echo "c4x2 exit $?"; tail -c 1500 gpurun_out/r2_bench_c4_2gpu.json
fi
