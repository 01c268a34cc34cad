#!/bin/bash
# round 6 mid-point: whole GPU suite, then the default bench (all legs)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6c_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r6c_bench_720p.json 2> gpurun_out/r6c_bench_720p.err
echo "bench exit $?" >> gpurun_out/r6c_gpu_tests.txt
cat gpurun_out/r6c_gpu_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_raft_f16")})
print("parity_timed_output", d.get("parity_timed_output"))
print("stress", d.get("stress"))
print("fallback", d.get("fallback"))
PY
