#!/bin/bash
# round 6: full GPU suite + default bench (with parity_timed_output) + the 3-branch streaming form once more with replay-to-replay detail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r6a_bench_720p.json 2> gpurun_out/r6a_bench_720p.err
echo "bench exit $?" >> gpurun_out/r6_gpu_tests.txt
O=gpurun_out/r6_hazards3.txt
: > $O
run() { echo "== $*" >> $O; timeout 900 python tools/check_hazards.py "$@" 2>&1 | grep -E "HAZARDS|Error|error|Traceback" | tail -3 >> $O; }
run stream 320 80 720 1280 0,1,2
run stream 320 80 720 1280 0,2
cat gpurun_out/r6_gpu_tests.txt
python - <<PY
import json
d = json.load(open("gpurun_out/r6a_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_raft_f16")})
print("parity_timed_output", d.get("parity_timed_output"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_us")} if d.get("roofline") else None)
for line in open("gpurun_out/r6_hazards3.txt"):
    if line.startswith("HAZARDS "):
        r=json.loads(line[8:]); print({k:(v if not isinstance(v,(list,dict)) or k=="replay_detail" else (len(v) if k=="graphs" else [m[:300] for m in v[:3]])) for k,v in r.items()})
    else: print(line.rstrip()[:300])
PY
tail -c 800 gpurun_out/r6a_bench_720p.err
