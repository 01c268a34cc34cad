#!/bin/bash
# round 6: old placement (chains inside a lane) with EVERY multi-tap stride-1 convolution of the generator on the register-staged kernel (no halo kernel in the windows at all):
# are kernels WITHOUT LDS-DMA in flight hit as well?
O=gpurun_out/r6_chain_victims.txt; : > $O
PP_CHAIN_IN_LANES=1 python tools/diag_replay_bytes.py 100 2 1 gen3x3=1 off=1 bb=1 2>&1 | grep -E "REPLAY_|Error" | tail -12 >> $O
cat $O
