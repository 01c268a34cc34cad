#!/bin/bash
# round 6: more hardware queues for the graph's branches -- does the shipped form stay clean, does the old placement get worse?
O=gpurun_out/r6_queues8.txt; : > $O
echo "== shipped form, DEBUG_HIP_FORCE_GRAPH_QUEUES=8" >> $O
DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python tools/diag_replay_bytes.py 100 2 2 2>&1 | grep -E "REPLAY_|Error" | tail -6 >> $O
echo "== old placement (PP_CHAIN_IN_LANES=1), DEBUG_HIP_FORCE_GRAPH_QUEUES=8" >> $O
PP_CHAIN_IN_LANES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python tools/diag_replay_bytes.py 100 2 1 2>&1 | grep -E "REPLAY_|Error" | tail -12 >> $O
echo "== old placement, DEBUG_HIP_FORCE_GRAPH_QUEUES=1" >> $O
PP_CHAIN_IN_LANES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 python tools/diag_replay_bytes.py 100 2 1 2>&1 | grep -E "REPLAY_|Error" | tail -12 >> $O
cat $O
