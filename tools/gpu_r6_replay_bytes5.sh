#!/bin/bash
O=gpurun_out/r6_replay_bytes5.txt; : > $O
python tools/diag_replay_bytes.py 100 2 2 stress 2>&1 | grep -E "REPLAY_|Error" | tail -8 >> $O
python tools/diag_replay_bytes.py 100 2 2 160 2>&1 | grep -E "REPLAY_|Error" | tail -8 >> $O
python tools/diag_replay_bytes.py 60 2 2 eager 2>&1 | grep -E "REPLAY_|Error" | tail -8 >> $O
cat $O
