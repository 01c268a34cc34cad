#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_setprio_ab_fp16.txt
: > $O
V=$(pwd)/build/setprio/libpropainter_hip.so
for r in 1 2; do
  echo "== fp16 base (round $r)" >> $O
  timeout 200 python tools/bench_conv.py --reps 30 --impls 0 --only "prop_|dec_|gen_qkv|gen_fc1|gen_softsplit|gen_enc_128|raft_gru|raft_convc2" 2>&1 | grep -E "^[a-z_0-9]+ +0 " >> $O
  echo "== fp16 setprio (round $r)" >> $O
  PP_LIB_PATH=$V timeout 200 python tools/bench_conv.py --reps 30 --impls 0 --only "prop_|dec_|gen_qkv|gen_fc1|gen_softsplit|gen_enc_128|raft_gru|raft_convc2" 2>&1 | grep -E "^[a-z_0-9]+ +0 " >> $O
done
cat $O
