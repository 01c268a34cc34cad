#!/bin/bash
# round 6: stand-alone ordering test of a multi-branch hipGraph shaped like the stage-pipelined streaming graph (tools/diag_graph_order.py)
O=${PP_GO_OUT:-gpurun_out/r6_graph_order.txt}; : > $O
run() { echo "== $*" >> $O; env $1 python tools/diag_graph_order.py ${@:2} 2>&1 | grep -E "GRAPH_ORDER|Error|error" | tail -3 >> $O; }
if [ -z "$PP_GO_SET2" ]; then
run PP_NOP=1 --replays 60 --mb 32
run PP_NOP=1 --replays 60 --mb 32 --alloc
run PP_NOP=1 --replays 30 --mb 128
run PP_NOP=1 --replays 30 --mb 128 --alloc
run PP_NOP=1 --replays 60 --mb 8 --kernels 200,40,10,150,10
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 --replays 30 --mb 32 --alloc
else
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 --replays 30 --mb 32
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 --replays 30 --mb 32 --alloc
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2 --replays 30 --mb 32 --alloc
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3 --replays 30 --mb 32 --alloc
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4 --replays 30 --mb 32 --alloc
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 --replays 30 --mb 32 --alloc --stages 0,2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 --replays 30 --mb 32 --alloc --stages ""
run PP_NOP=1 --replays 200 --mb 16 --alloc
fi
cat $O
