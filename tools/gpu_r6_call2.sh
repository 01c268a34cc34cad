#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_halo8.sh > gpurun_out/r6_halo8_stdout.txt 2>&1
bash tools/gpu_r6_hazards.sh > gpurun_out/r6_hazards_stdout.txt 2>&1
tail -c 6000 gpurun_out/r6_halo8.txt
tail -c 5000 gpurun_out/r6_hazards_stdout.txt
