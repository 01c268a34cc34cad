#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_stream_single.txt
: > $O
PP_SG_DEBUG=1 timeout 300 python -X faulthandler tools/diag_stream2.py 24 single chained 2>&1 | grep -E "STREAM_DIAG2|Fatal|File \"/root/repo|Error" | head -40 >> $O
cat $O
