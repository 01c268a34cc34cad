#!/bin/bash
# round 6: the ping-pong halo kernel (conv_halo8.h, impl 82 / 83) -- bit-identity tests, then kbench / bench_split A/B against the
# 128-pixel kernel for the shipped form and the A/B libraries under build/ (tools/build_h8_variant.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_halo8.txt
: > $O
timeout 1500 python -m pytest tests/test_halo8_gpu.py -x -q 2>&1 | tail -6 >> $O
K=build/kbench
for V in main h8_prio0 h8_prio2; do
  if [ $V = main ]; then LP=propainter_amd/lib; else LP=build/$V; fi
  [ -f $LP/libpropainter_hip.so ] || continue
  echo "######## library $V" >> $O
  kb() { echo "== $*" >> $O; LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH timeout 300 $K "$@" 2>&1 | grep "impl" | tail -4 >> $O; }
  kb conv 16 90 160 3 3 256 256 --impls 71,82,71,82 --act 1 --reps 40
  kb conv 16 90 160 1 5 256 128,128 --impls 71,82,71,82 --act 3 --late zr --reps 40
  kb conv 16 90 160 5 1 128 128,128 --impls 71,82,71,82 --act 4 --late h --reps 40
  kb conv 16 90 160 3 3 128 128 --impls 71,82,71,82 --act 1 --reps 40
  kb conv 16 90 160 3 3 64 128 --impls 72,83,72,83 --act 1 --reps 40
  kb conv 18 180 320 3 3 128 128,128 --impls 71,82,71,82 --act 0 --res --reps 30
  echo "== split-plane layers (tools/bench_split.py, RAFT 720p chunk), library $V" >> $O
  PP_LIB_PATH=$PWD/$LP/libpropainter_hip.so timeout 900 python tools/bench_split.py --reps 20 --only "gru|convc2|fh1|enc_3x3_128" 2>&1 | grep -v "^layer\|AMD\|Instinct" | tail -24 >> $O
done
cat $O | cut -c1-200
