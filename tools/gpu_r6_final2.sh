#!/bin/bash
# round 6, after the lane fix: whole GPU suite, then the bare default bench (roofline.traffic from the committed r6z PMC profile: same csrc digest)
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6z_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r6z_bench_720p.json 2> gpurun_out/r6z_bench_720p.err
echo "bench exit $?" >> gpurun_out/r6z_gpu_tests.txt
cat gpurun_out/r6z_gpu_tests.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6z_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_raft_f16")})
r = d["roofline"]; print({k: r[k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_us")})
print("parity_timed_output", {k: v for k, v in d["parity_timed_output"].items() if k != "what"})
print("stress", d["stress"]["value"], d["stress"]["kernels_ms"])
PY
