#!/bin/bash
# One GPU session of round 3: parity suite (prints kept), smoke, headline bench.  Outputs under gpurun_out/.
TAG=${1:-r3}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=8 $PYTEST_ARGS > gpurun_out/${TAG}_pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
  grep -E "HEADLINE_PARITY|PARITY |EPE|passed|failed|rror|exit|s call|s setup" gpurun_out/${TAG}_pytest_gpu.log | grep -v SPLIT_PARITY | tail -60
fi
if [ -z "$SKIP_SMOKE" ]; then
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; grep smoke gpurun_out/${TAG}_smoke.log | tail -4
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 1200 python bench.py --steps ${STEPS:-3} --warmup 1 $BENCH_ARGS > gpurun_out/${TAG}_bench_720.json 2> gpurun_out/${TAG}_bench_720.err
  echo "720p exit $?"; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/${TAG}_bench_720.json'))
    print({k:d.get(k) for k in ('value','value_raft_f16','ms_per_step','dtype','roofline','stages_ms','cpu_baseline','parity','raft_precisions','memory')})
    for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms']):
        print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
except Exception as e:
    print('bench parse failed', e)
PY
  tail -5 gpurun_out/${TAG}_bench_720.err
fi
