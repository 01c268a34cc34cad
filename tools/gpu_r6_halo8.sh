#!/bin/bash
# round 6: the ping-pong halo kernel (conv_halo8.h, impl 82 / 83) -- bit-identity tests, then interleaved kbench A/B against the 128-pixel kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_halo8.txt
: > $O
timeout 1500 python -m pytest tests/test_halo8_gpu.py -x -q 2>&1 | tail -15 >> $O
K=build/kbench
kb() { echo "== $*" >> $O; timeout 300 $K "$@" 2>&1 | grep -v "^$" | tail -6 >> $O; }
kb conv 16 90 160 3 3 256 256 --impls 71,82,71,82 --act 1 --reps 40
kb conv 16 90 160 1 5 256 128,128 --impls 71,82,71,82 --act 3 --late zr --reps 40
kb conv 16 90 160 5 1 128 128,128 --impls 71,82,71,82 --act 4 --late h --reps 40
kb conv 16 90 160 3 3 128 128 --impls 71,82,71,82 --act 1 --reps 40
kb conv 16 90 160 3 3 64 128 --impls 72,83,72,83 --act 1 --reps 40
kb conv 16 90 160 3 3 192 256 --impls 72,83,72,83 --act 1 --reps 40
kb conv 2 360 640 3 3 128 128 --impls 71,82,71,82 --act 1 --reps 30
kb conv 18 180 320 3 3 128 128,128 --impls 71,82,71,82 --act 0 --res --reps 30
kb conv 16 90 160 3 3 256 256 --impls 71,82 --act 1 --reps 40 --zero
echo "== split-plane layers (tools/bench_split.py, RAFT 720p chunk)" >> $O
timeout 900 python tools/bench_split.py --reps 20 --only "gru|convc2|convf2|convm|fh1|enc_3x3" 2>&1 | tail -40 >> $O
cat $O | cut -c1-250
