#!/bin/bash
# round 6: the old placement (chains inside a lane) with variant libraries whose launches ask for enough dynamic LDS that LDS-using blocks never share a CU:
# all kernels (build/excl), or one kernel family only (build/excl_dcn, excl_halo, excl_v2, excl_rest)
O=${PP_CE_OUT:-gpurun_out/r6_chain_excl.txt}; : > $O
for v in ${PP_CE_VARIANTS:-excl}; do
echo "== PP_CHAIN_IN_LANES=1, variant library build/$v" >> $O
PP_CHAIN_IN_LANES=1 PP_LIB_PATH=build/$v/libpropainter_hip.so python tools/diag_replay_bytes.py ${PP_CE_REPLAYS:-100} 2 1 2>&1 | grep -E "REPLAY_|Error|error" | tail -12 >> $O
done
cat $O
