#!/bin/bash
# round 6: the stage-pipelined single graph at config 4 with only RAFT on a branch (the rule of profiles/r6_replay_bytes.txt) vs RAFT + image propagation
O=gpurun_out/r6_stream_rule.txt; : > $O
for st in 0 0,2; do echo "== stages $st, 16 replays" >> $O
PP_HZ_REPLAYS=16 python tools/check_hazards.py stream 320 80 720 1280 $st 2>&1 | grep HAZARDS | python -c "
import sys,json
d=json.loads(sys.stdin.read()[8:]); r=d['replay_detail']
print('replays_equal_eager', sum(d['replays_equal_eager']), 'of', len(d['replays_equal_eager']), 'bytes_differing', r['bytes_differing_from_eager'], 'max_abs', r['max_abs_vs_eager'], 'replay_ms', r.get('replay_ms_under_the_recorder'), 'findings', {k:d[k] for k in d if k.endswith('_count')})" >> $O 2>&1
done
cat $O
