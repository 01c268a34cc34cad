#!/bin/bash
# BASELINE configs 2, 4, 5 on ONE GPU (TAG=r6z ...) (timed default: fp16 stages + f16x3 RAFT; config 5 also with fp16 RAFT).
mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline --no-precisions --no-stress --no-configs > gpurun_out/${TAG:-r6z}_bench_config_$name.json 2> gpurun_out/${TAG:-r6z}_bench_config_$name.err; echo "$name exit $?"; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/${TAG:-r6z}_bench_config_$name.json'))
    print('$name', round(d['value'],2), 'frames/s', round(d['ms_per_step'],1), 'ms', d['config']['workload'], '| RAFT', d['config']['raft_dtype'], '|', {k: round(v,1) for k,v in d['memory'].items() if k!='note'}, '|', d['submission'][:40])
    if d.get('roofline'): print('   roofline', d['roofline']['kernel'], round(d['roofline']['achieved'],1), round(d['roofline']['frac'],3))
except Exception as e:
    print('$name parse failed', e); print(open('gpurun_out/${TAG:-r6z}_bench_config_$name.err').read()[-1500:])
PY
}
run c4_720p_320f --sharded --frames 320 --steps 2 --warmup 1
run c5_1080p_160f_f16x3 --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20 --steps 2 --warmup 1
