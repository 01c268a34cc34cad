#!/bin/bash
# round 6: rate of deviating replays of the whole-pass graph, by lane count, by kernel build, graph vs eager (profiles/r6_replay_bytes.txt)
O=${PP_RB_OUT:-gpurun_out/r6_replay_bytes.txt}; : > $O
if [ -z "$PP_RB_SET2" ]; then
python tools/diag_replay_bytes.py 40 2 2 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 40 1 1 2>&1 | grep REPLAY_ >> $O
PP_LIB_PATH=build/dcn_old/libpropainter_hip.so python tools/diag_replay_bytes.py 40 2 2 2>&1 | grep -E "REPLAY_|Error" >> $O
else
python tools/diag_replay_bytes.py 40 2 1 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 40 1 2 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 30 2 2 eager 2>&1 | grep REPLAY_ >> $O
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 python tools/diag_replay_bytes.py 40 2 2 2>&1 | grep REPLAY_ >> $O
F="--no-cpu-baseline --no-profile --no-precisions --no-configs --no-stress --steps 5 --warmup 2"
for cfg in "1 1" "2 2" "1 2" "2 1" "1 1" "2 2"; do set -- $cfg; echo "== bench window_streams $1 raft_streams $2" >> $O
  python bench.py $F --window-streams $1 --raft-streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity_timed_output',{}).get('max_abs'))" >> $O 2>&1; done
fi
cat $O
