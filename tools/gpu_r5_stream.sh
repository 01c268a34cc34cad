#!/bin/bash
# round 5: first-pass statistics of the streaming schedule per form and per HIP-runtime variant (tools/diag_stream2.py)
mkdir -p gpurun_out
O=gpurun_out/r5_streaming_race.txt
: > $O
timeout 400 python tools/diag_stream2.py 16 concurrent concurrent_d2d concurrent_fence single chained 2>&1 | grep -E "STREAM_DIAG2|Error|error|Traceback" >> $O
for v in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=0; do
  env $v timeout 240 python tools/diag_stream2.py 16 concurrent 2>&1 | grep -E "STREAM_DIAG2|Error|error|Traceback" >> $O
done
cat $O
