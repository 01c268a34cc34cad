#!/bin/bash
# round 6: the timed output against the 80-frame golden (GPU test + bench line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stress_gpu.py -x -q -s -k "config3_timed" 2>&1 | grep -E "STRESS_E2E|passed|failed|Error|assert" | tail -8 > gpurun_out/r6_timed_golden.txt
timeout 900 python bench.py --no-stress --no-configs --no-precisions > gpurun_out/r6b_bench_720p.json 2> gpurun_out/r6b_bench_720p.err
echo "bench exit $?" >> gpurun_out/r6_timed_golden.txt
cat gpurun_out/r6_timed_golden.txt
python - <<PY
import json
d = json.load(open("gpurun_out/r6b_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step")})
print("parity_timed_output", d.get("parity_timed_output"))
print("parity", d.get("parity"))
PY
tail -c 600 gpurun_out/r6b_bench_720p.err
