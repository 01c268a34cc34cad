#!/bin/bash
# round-4 final measurements of commit beb53a7: default bench, rocprofv3 kernel trace + FETCH / WRITE PMC passes, MFMA-busy, LDS conflicts, configs 2 / 4 / 5
export COMMIT=beb53a7 COMMIT_TIME=1790195688 RAFT_DTYPE=f16x3
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r4x_bench_720p.json 2> gpurun_out/r4x_bench_720p.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4x_bench_720p.json'))
print({k:d.get(k) for k in ('value','ms_per_step','value_raft_f16','parity')}); print(d['roofline']['frac'], d['roofline']['achieved'], d['memory'])
PY
bash tools/gpu_profile.sh r4y 2>&1 | tail -45
bash tools/gpu_mfma_pmc.sh r4y 2>&1 | tail -14
bash tools/gpu_lds_pmc.sh r4y 2>&1 | tail -14
bash tools/gpu_configs_r4.sh 2>&1 | tail -12
du -sh gpurun_out
