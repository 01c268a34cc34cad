#!/bin/bash
# same-box, pass-level A/B of the shipped s_setprio switches: in-tree library (split + attention clusters ON) vs build/noprio (both OFF)
mkdir -p gpurun_out
O=gpurun_out/r5_setprio_pass_ab3.txt
: > $O
one() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precisions --no-stress --no-configs 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']; print('$1', round(d['ms_per_step'],1), 'ms;', {n: round(k[n]['ms'],1) for n in ('conv_gemm_f16x3','conv_gemm_f16','sparse_window_attention','conv_gemm_dcn','corr_lookup_otf_split')})"
}
for r in 1 2; do
  one "shipped (split + attention priority)" >> $O
  PP_LIB_PATH=$(pwd)/build/noprio/libpropainter_hip.so one "no priority (round-4 schedule)     " >> $O
done
cat $O
