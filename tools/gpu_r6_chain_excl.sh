#!/bin/bash
# round 6: the old placement (chains inside a lane) with a variant library whose every launch asks for enough dynamic LDS that LDS-using blocks never share a CU
O=gpurun_out/r6_chain_excl.txt; : > $O
echo "== PP_CHAIN_IN_LANES=1, variant library (exclusive LDS)" >> $O
PP_CHAIN_IN_LANES=1 PP_LIB_PATH=build/excl/libpropainter_hip.so python tools/diag_replay_bytes.py 100 2 1 2>&1 | grep -E "REPLAY_|Error|error" | tail -12 >> $O
cat $O
