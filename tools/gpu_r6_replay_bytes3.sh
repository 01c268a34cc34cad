#!/bin/bash
O=gpurun_out/r6_replay_bytes3.txt; : > $O
python tools/diag_replay_bytes.py 150 2 2 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 150 1 2 2>&1 | grep REPLAY_ >> $O
F="--no-cpu-baseline --no-profile --no-precisions --no-configs --no-stress --steps 5 --warmup 2"
echo "== bench default" >> $O
python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity_timed_output',{}).get('max_abs'))" >> $O 2>&1
cat $O
