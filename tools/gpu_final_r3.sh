#!/bin/bash
# Final GPU session of round 3: full parity suite + smoke + default bench (tools/gpu_round3.sh), clock / power telemetry of a plain bench,
# BASELINE configs 2 and 5 on one GPU with the final code.
TAG=${1:-r3f}
STEPS=3 bash tools/gpu_round3.sh $TAG
timeout 300 python tools/gpu_telemetry.py gpurun_out/${TAG}_telemetry_bench.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-precisions --no-profile > gpurun_out/${TAG}_telemetry_bench.log 2>&1
echo "telemetry exit $?"; python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_telemetry_bench.json'))['summary']
print({k:d[k] for k in ('card','cards_sampled','mean_busy_pct_per_card','while_busy')})
PY
timeout 300 python bench.py --frames 80 --height 240 --width 432 --steps 5 --warmup 2 --no-cpu-baseline --no-precisions --no-profile > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
echo "c2 exit $?"; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_c2.json')); print('c2', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20 --steps 2 --warmup 1 --no-cpu-baseline --no-precisions --no-profile > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
echo "c5 exit $?"; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); print('c5', d['value'], d['ms_per_step'], d.get('memory'))"; tail -2 gpurun_out/${TAG}_bench_c5.err
