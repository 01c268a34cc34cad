#!/bin/bash
# rocprofv3 passes of the headline bench command (one eager step): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE PMC
# passes (separate runs, as the MI355X guide prescribes).  Summaries -> gpurun_out/<tag>_*.{md,json}.
TAG=${1:-r2}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --single-pass --window-streams 1 --raft-streams 1 --no-cpu-baseline $BENCH_ARGS"   # windows serialised: per-launch durations are not overlapped
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace --output-format csv -- ${CMD/--single-pass/--steady-pass} > $R/gpurun_out/prof_trace.log 2>&1
echo "trace exit $?"
if [ -z "$SKIP_PMC" ]; then
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1
  echo "pmc fetch exit $?"
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- $CMD > $R/gpurun_out/pmc_write.log 2>&1
  echo "pmc write exit $?"
fi
cd $R
python tools/rocprof_summary.py stats gpurun_out/prof_trace gpurun_out/${TAG}_rocprof_kernel_stats_720p.md | head -60
# (the snapshot on the GPU box has no .git: the caller passes the commit it pushed as COMMIT / COMMIT_TIME)
[ -z "$SKIP_PMC" ] && python tools/rocprof_summary.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${TAG}_hbm_traffic_720p.json "${COMMIT:-unknown}" "${COMMIT_TIME:-0}" "${RAFT_DTYPE:-f16x3}" | head -40
find gpurun_out/prof_trace gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +8M -delete 2>/dev/null
du -sh gpurun_out
