#!/bin/bash
# MFMA utilisation + effective clock per kernel family of one eager pass (north_star: "MFMA utilisation against peak").
# One rocprofv3 --pmc pass WITH --kernel-trace (counters and durations of the very same dispatches; --pmc plus trace domains other
# than the kernel trace are refused on this pool):
#   SQ_VALU_MFMA_BUSY_CYCLES   cycles the matrix pipes are busy, summed over SIMDs (a 16x16x32 f16 MFMA = 16 cycles = the pipe's peak rate)
#   SQ_INSTS_VALU_MFMA_MOPS_F16  fp16 matrix operations issued
#   SQ_BUSY_CYCLES / SQ_WAVE_CYCLES   shader-engine busy time / wave residency (quad-cycles)
#   GRBM_GUI_ACTIVE            cycles the GPU was active during the dispatch -> effective clock = GRBM_GUI_ACTIVE / duration
# Summary -> gpurun_out/<tag>_mfma_util_720p.json / .txt  (tools/rocprof_summary.py mfma)
TAG=${1:-r3}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --single-pass --window-streams 1 --raft-streams 1 --no-cpu-baseline $BENCH_ARGS"
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma --output-format csv -- $CMD > $R/gpurun_out/pmc_mfma.log 2>&1
echo "pmc mfma exit $?"
cd $R
python tools/rocprof_summary.py mfma gpurun_out/pmc_mfma gpurun_out/${TAG}_mfma_util_720p.json "${COMMIT:-unknown}" | tee gpurun_out/${TAG}_mfma_util_720p.txt
find gpurun_out/pmc_mfma -name "*.csv" -size +8M -delete 2>/dev/null
