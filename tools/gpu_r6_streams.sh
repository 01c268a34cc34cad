#!/bin/bash
# (profiles/r6_stream_counts.txt)
# round 6: window / RAFT stream counts re-measured on the final kernels (same box, back to back)
F="--no-cpu-baseline --no-profile --no-precisions --no-configs --no-stress --steps 5 --warmup 2"
for cfg in "2 2" "3 2" "4 2" "2 3" "2 4" "3 3" "2 2"; do
  set -- $cfg
  echo "== window_streams $1 raft_streams $2" >> gpurun_out/r6_streams.txt
  python bench.py $F --window-streams $1 --raft-streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity_timed_output',{}).get('max_abs'))" >> gpurun_out/r6_streams.txt 2>&1
done
cat gpurun_out/r6_streams.txt
