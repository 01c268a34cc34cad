#!/bin/bash
# Bench only: 720p headline with per-step / per-stage timing.  Usage: gpu_bench.sh [bench args]
mkdir -p gpurun_out
timeout 900 python bench.py "$@" > gpurun_out/bench_720.json 2> gpurun_out/bench_720.err
echo "720p exit $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_720.json'))
print({k:d.get(k) for k in ('value','ms_per_step','step_ms','host_submit_ms','roofline','stages_ms','cpu_baseline')})
for k,v in sorted((d.get('kernels') or {}).items(), key=lambda kv:-kv[1]['ms']):
    print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
PY
tail -3 gpurun_out/bench_720.err
