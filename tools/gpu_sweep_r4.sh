#!/bin/bash
# stream-count sweep of the whole-pass graph on the round-4 code (one box, back to back)
for cfg in "2 3" "3 3" "2 4" "2 2" "2 3"; do set -- $cfg
  timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-precisions --no-profile --window-streams $1 --raft-streams $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SWEEP window_streams $1 raft_streams $2:', round(d['ms_per_step'], 1), 'ms', round(d['value'], 2), 'frames/s', {k: round(v, 1) for k, v in d['memory'].items() if k != 'note'})"
done
