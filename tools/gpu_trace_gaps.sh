#!/bin/bash
# rocprofv3 kernel trace of hipGraph replays of the headline pass -> idle time between kernels inside a replay (tools/trace_gaps.py)
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_replay --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precisions --no-profile > $R/gpurun_out/trace_replay.log 2>&1
echo "trace exit $?"
cd $R
python tools/trace_gaps.py gpurun_out/trace_replay --window-ms 1400 | tee gpurun_out/r3z_replay_gaps.json
tail -2 gpurun_out/trace_replay.log | cut -c1-300
find gpurun_out/trace_replay -name "*.csv" -size +1M -delete 2>/dev/null
