#!/bin/bash
# round 6, config 5 through bench.py itself (the form whose replay_consistency leg saw 1 of 16 replays deviate): N more replays with the default
# lanes and with no lanes at all.
mkdir -p gpurun_out
O=gpurun_out/r6_c5_bench_replays.txt
: > $O
for cfg in "2 2" "1 1"; do
  set -- $cfg
  timeout 800 python bench.py --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20 --steps 2 --warmup 1 --no-cpu-baseline --no-precisions --no-stress --no-configs \
     --window-streams $1 --raft-streams $2 --replay-checks ${N:-48} 2> gpurun_out/c5b.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('window_streams $1 raft_streams $2', round(d['value'], 2), 'frames/s', d['replay_consistency'])" >> $O
done
cat $O
