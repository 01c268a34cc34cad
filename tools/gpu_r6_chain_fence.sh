#!/bin/bash
# round 6: the old placement (single windows' propagation chains inside their lane) with and without a cache-maintenance kernel behind EVERY engine launch
O=gpurun_out/r6_chain_fence.txt; : > $O
echo "== PP_CHAIN_IN_LANES=1 (control: the defect as shipped in rounds 2-5)" >> $O
PP_CHAIN_IN_LANES=1 python tools/diag_replay_bytes.py 80 2 1 2>&1 | grep REPLAY_ >> $O
echo "== PP_CHAIN_IN_LANES=1 PP_FENCE_EVERY_LAUNCH=3 (write-back + invalidate of every XCD's L2 behind every engine launch)" >> $O
PP_CHAIN_IN_LANES=1 PP_FENCE_EVERY_LAUNCH=3 python tools/diag_replay_bytes.py 80 2 1 2>&1 | grep REPLAY_ >> $O
cat $O
