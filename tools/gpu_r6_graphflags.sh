#!/bin/bash
# round 6: is the run-to-run deviation of the three-branch single graph (720x1280x320, stages 0,1,2) a property of how the HIP runtime
# executes a multi-branch graph?  Same program, runtime knobs varied (DEBUG_* flags of libamdhip64: strings | grep GRAPH).
O=${PP_GF_OUT:-gpurun_out/r6_graphflags.txt}; : > $O
run() { echo "== $1" >> $O; env $1 python tools/check_hazards.py stream 320 80 720 1280 0,1,2 2>&1 | grep HAZARDS | python -c "
import sys,json
d=json.loads(sys.stdin.read()[8:]); r=d['replay_detail']
print('replays_equal_eager', d['replays_equal_eager'], 'bytes_differing', r['bytes_differing_from_eager'], 'max_abs', r['max_abs_vs_eager'], 'replay_ms', r.get('replay_ms_under_the_recorder'), 'replay_i==replay_0', r['replay_i_equals_replay_0'], 'seconds', d['seconds'], 'findings', {k:d[k] for k in d if k.endswith('_count')})" >> $O 2>&1; }
for v in ${PP_GF_RUNS:-PP_NOP=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=4}; do run "$v"; done
cat $O
