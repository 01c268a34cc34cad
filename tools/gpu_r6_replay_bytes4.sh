#!/bin/bash
O=gpurun_out/r6_replay_bytes4.txt; : > $O
python tools/diag_replay_bytes.py 300 2 2 80 240 432 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 300 1 1 80 240 432 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 300 1 2 80 240 432 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 300 2 1 80 240 432 2>&1 | grep REPLAY_ >> $O
python tools/diag_replay_bytes.py 60 2 2 40 1080 1920 2>&1 | grep REPLAY_ >> $O
cat $O
