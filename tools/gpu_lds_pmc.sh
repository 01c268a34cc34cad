#!/bin/bash
# LDS bank-conflict counters of one eager pass (rocprofv3 --pmc, own run): SQ_LDS_BANK_CONFLICT = extra LDS cycles,
# SQ_LDS_IDX_ACTIVE = all LDS-array cycles.  Summary -> gpurun_out/<tag>_lds_conflicts.json
TAG=${1:-r2}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --single-pass --window-streams 1 --raft-streams 1 --no-cpu-baseline $BENCH_ARGS"
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_lds --output-format csv -- $CMD > $R/gpurun_out/pmc_lds.log 2>&1
echo "pmc lds exit $?"
cd $R
python tools/rocprof_summary.py counters gpurun_out/pmc_lds gpurun_out/${TAG}_lds_conflicts.json SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE | tee gpurun_out/${TAG}_lds_conflicts.txt
rm -rf gpurun_out/pmc_lds
