#!/bin/bash
# round 4, call 1: the new parity tests + a short bench of the volume-free split-plane correlation
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s \
  -k "split_plane or timed_shape or image_propagation or evaluation_protocol or raft" > gpurun_out/r4a_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4a_pytest.log
grep -E "PARITY|STRESS|passed|failed|Error|error|exit" gpurun_out/r4a_pytest.log | tail -80
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precisions > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r4a_bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r4a_bench.json'))
    print({k:d.get(k) for k in ('value','ms_per_step','roofline','stages_ms','memory')})
    for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms']):
        print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
except Exception as e:
    print('bench parse failed', e)
PY
