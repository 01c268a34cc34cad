#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_split_plane_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "stress or overflow or volume_free" > gpurun_out/r4b_pytest.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r4b_pytest.log | tail -5
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precisions > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r4b_bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r4b_bench.json'))
    print({k:d.get(k) for k in ('value','ms_per_step','memory')}); print(d['stages_ms'])
    for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:12]:
        print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
except Exception as e:
    print('bench parse failed', e)
PY
