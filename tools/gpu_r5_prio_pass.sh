#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_setprio_pass_ab2.txt
: > $O
one() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precisions --no-stress --no-configs 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']; print('$1', round(d['ms_per_step'],1), 'ms;', {n: round(k[n]['ms'],1) for n in ('conv_gemm_f16x3','conv_gemm_f16','sparse_window_attention','conv_gemm_dcn','corr_lookup_otf_split')})"
}
for r in 1 2; do
  one "base(split=1,attn=1)" >> $O
  PP_LIB_PATH=$(pwd)/build/prio_early/libpropainter_hip.so one "split_early        " >> $O
done
cat $O
