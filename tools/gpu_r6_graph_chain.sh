#!/bin/bash
O=gpurun_out/r6_graph_chain.txt; : > $O
run() { echo "== $*" >> $O; python tools/diag_graph_chain.py "$@" 2>&1 | grep -E "GRAPH_CHAIN|Error" | tail -2 >> $O; }
run --replays 200
run --replays 200 --both-chains
run --replays 200 --small 262144 --chain 100
run --replays 100 --chain 600 --small 16384 --big-mb 256 --big-kernels 30
run --replays 200 --small 1024 --chain 200 --big-mb 16 --big-kernels 4
cat $O
