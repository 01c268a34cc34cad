#!/bin/bash
# end of round 6: whole GPU suite, the bare default bench, config 5 with 32 replay checks (all at the no-forked-branches default)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6f_gpu_tests.txt
cat gpurun_out/r6f_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r6f_bench_720p.json 2> gpurun_out/r6f_bench_720p.err
echo "bench exit $?" | tee -a gpurun_out/r6f_gpu_tests.txt
timeout 700 python bench.py --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20 --steps 2 --warmup 1 --no-cpu-baseline --no-precisions --no-stress --no-configs --replay-checks 32 > gpurun_out/r6f_bench_config_c5_1080p_160f_f16x3.json 2> gpurun_out/r6f_c5.err
echo "c5 exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6f_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_raft_f16")}, d["submission"])
r = d["roofline"]; print({k: r[k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_us")})
print("parity_timed_output", {k: v for k, v in d["parity_timed_output"].items() if k != "what"})
print("replay_consistency", d["replay_consistency"]); print("stress", d["stress"]["value"])
c = json.load(open("gpurun_out/r6f_bench_config_c5_1080p_160f_f16x3.json"))
print("c5", c["value"], c["replay_consistency"], c["submission"])
PY
