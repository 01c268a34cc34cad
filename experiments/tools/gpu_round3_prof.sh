#!/bin/bash
# Profiling session of round 3 (one gpurun call): per-layer bench detail, rocprofv3 kernel stats + HBM traffic PMC passes, MFMA
# utilisation PMC pass, clock / power telemetry during a plain bench.  COMMIT / COMMIT_TIME = the commit that was pushed.
TAG=${1:-r3}
export COMMIT COMMIT_TIME RAFT_DTYPE=${RAFT_DTYPE:-f16x3}
mkdir -p gpurun_out
timeout 600 python bench.py --steps 2 --warmup 1 --detail --no-cpu-baseline --no-precisions > gpurun_out/${TAG}_bench_detail.json 2> gpurun_out/${TAG}_bench_detail.err
echo "detail exit $?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_detail.json'))
print(d['value'], d['ms_per_step'])
rows=sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])
for k,v in rows[:70]:
    print(f"{v['ms']:8.2f} ms n={v['launches']:5d} avg={v['avg_us']:8.1f}us TF={v['tflops']:7.1f} GB/s={v['gbs']:7.1f}  {k}")
PY
bash tools/gpu_profile.sh ${TAG} 2>&1 | tail -75
bash tools/gpu_mfma_pmc.sh ${TAG} 2>&1 | tail -30
timeout 400 python tools/gpu_telemetry.py gpurun_out/${TAG}_telemetry_bench.json -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-precisions --no-profile > gpurun_out/${TAG}_telemetry_bench.log 2>&1
echo "telemetry exit $?"; tail -40 gpurun_out/${TAG}_telemetry_bench.log | head -60
