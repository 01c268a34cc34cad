#!/bin/bash
# round-4 end-of-round measurements of commit 819b760: rocprofv3 kernel trace (steady pass) + FETCH / WRITE PMC passes, MFMA-busy, LDS conflicts, configs 2 / 4 / 5
export COMMIT=819b760 COMMIT_TIME=1790200669 RAFT_DTYPE=f16x3
mkdir -p gpurun_out
bash tools/gpu_profile.sh r4z 2>&1 | tail -48
bash tools/gpu_mfma_pmc.sh r4z 2>&1 | tail -14
bash tools/gpu_lds_pmc.sh r4z 2>&1 | tail -14
sed -i 's/r4_bench_/r4z_bench_/g' tools/gpu_configs_r4.sh
bash tools/gpu_configs_r4.sh 2>&1 | tail -12
du -sh gpurun_out
