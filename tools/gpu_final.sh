#!/bin/bash
# end-of-round measurements (TAG=r6z ...): the bare default bench (headline + roofline + cpu_baseline + parity + fallback / configs / stress legs),
# rocprofv3 kernel trace of a steady pass + FETCH / WRITE PMC passes (stamped with the commit and the csrc digest), MFMA-busy, LDS conflicts,
# BASELINE config 4 as whole-pass graph vs stage-pipelined streaming graph vs chained segment graphs.
export COMMIT=${COMMIT:-unknown} COMMIT_TIME=${COMMIT_TIME:-0} RAFT_DTYPE=f16x3
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG:-r6z}_bench_720p.json 2> gpurun_out/${TAG:-r6z}_bench_720p.err
echo "bench exit $?"; tail -c 600 gpurun_out/${TAG:-r6z}_bench_720p.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/${TAG:-r6z}_bench_720p.json"))
print({k: d[k] for k in ("value", "ms_per_step", "value_raft_f16")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_us")} if d.get("roofline") else None)
print("parity", d.get("parity")); print("parity_refs", d.get("parity_windows_with_reference_frames"))
print("fallback", d.get("fallback")); print("configs", d.get("configs")); print("stress", d.get("stress"))
print("memory", {k: v for k, v in d["memory"].items() if k != "note"})
PY
bash tools/gpu_profile.sh ${TAG:-r6z} 2>&1 | tail -30
bash tools/gpu_mfma_pmc.sh ${TAG:-r6z} 2>&1 | tail -12
bash tools/gpu_lds_pmc.sh ${TAG:-r6z} 2>&1 | tail -14
timeout 600 python tools/bench_streaming.py --steps 2 > gpurun_out/${TAG:-r6z}_streaming_config4.json 2> gpurun_out/${TAG:-r6z}_streaming_config4.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/${TAG:-r6z}_streaming_config4.json"))
print({k: v for k, v in d.items() if "ms_per" in str(v) or "equal" in k or "GB" in k or "speedup" in k})
PY
du -sh gpurun_out
