#!/bin/bash
# round 6, config 5 (1080x1920x160, --subvideo_length 20): the default bench's replay_consistency leg saw 1 of 16 replays deviate there
# (profiles/r6zz_bench_config_c5_1080p_160f_f16x3.json).  Deviating replays per lane configuration.
mkdir -p gpurun_out
O=gpurun_out/r6_c5_replays.txt
: > $O
for cfg in "2 2" "1 2" "2 1" "1 1"; do
  set -- $cfg
  timeout 700 python tools/diag_replay_bytes.py ${N:-40} $1 $2 160 1080 1920 sub=20 2>&1 | grep "REPLAY_" >> $O
  echo "--- window_streams $1 raft_streams $2 exit $?" >> $O
done
cat $O
