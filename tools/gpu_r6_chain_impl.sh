#!/bin/bash
# round 6: old placement (chains inside a lane), the CHAIN's convolutions switched to the register-staged kernel (impl 1: no LDS-DMA, small LDS) -- offender or victim?
O=gpurun_out/r6_chain_impl.txt; : > $O
echo "== chain convolutions (off*, bb*) register-staged" >> $O
PP_CHAIN_IN_LANES=1 python tools/diag_replay_bytes.py 120 2 1 off=1 bb=1 2>&1 | grep -E "REPLAY_|Error" | tail -10 >> $O
echo "== chain convolutions and the deformable convolution register-staged" >> $O
PP_CHAIN_IN_LANES=1 python tools/diag_replay_bytes.py 120 2 1 off=1 bb=1 dcn=1 2>&1 | grep -E "REPLAY_|Error" | tail -10 >> $O
cat $O
