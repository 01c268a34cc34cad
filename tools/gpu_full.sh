#!/bin/bash
# Full GPU check: test-suite, then the 720p bench (one timed step) with the per-kernel breakdown.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_720.json 2> gpurun_out/bench_720.err
echo "720p exit $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_720.json'))
print({k:d[k] for k in ('value','ms_per_step','roofline','stages_ms')})
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms']):
    print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
PY
tail -3 gpurun_out/bench_720.err
