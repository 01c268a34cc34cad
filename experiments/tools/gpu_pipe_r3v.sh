#!/bin/bash
# Round 3: the software-pipelined halo kernel (impl 116) against the shipped one (impl 70 = auto halo), fp16 via kbench, split-plane via bench_split.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=build/kbench
{
$K conv 16 90 160 3 3 256 256 --impls 70,116,70,116 --act 1 --reps 40
$K conv 16 90 160 1 5 256 128,128 --impls 70,116,70,116 --act 1 --reps 40
$K conv 16 90 160 5 1 128 128,128 --impls 70,116,70,116 --act 4 --late h --reps 40
$K conv 16 90 160 1 5 256 128,128 --impls 70,116,70,116 --act 3 --late zr --reps 40
$K conv 16 90 160 3 3 128 128 --impls 70,116,70,116 --act 1 --reps 40
$K conv 1 180 320 3 3 128 128,128 --impls 70,116,70,116 --act 0 --res --reps 60
$K conv 2 360 640 3 3 128 128 --impls 70,116,70,116 --act 1 --reps 40
$K conv 1 180 320 3 3 432 128 --impls 70,116,70,116 --act 0 --reps 60
$K conv 4 360 640 3 3 64 64 --impls 70,116,70,116 --act 1 --reps 40
$K conv 1 180 320 3 3 128 128,128,64 --impls 70,116,70,116 --act 1 --reps 60
} > gpurun_out/r3v_kbench_pipe.txt 2>&1
cat gpurun_out/r3v_kbench_pipe.txt
timeout 600 python tools/bench_split.py --reps 20 --only "c2_3x3|convf2|convm|gru|fh1|enc_3x3_" > gpurun_out/r3v_split_pipe.txt 2> gpurun_out/r3v_split_pipe.err
echo "sweep exit $?"; cat gpurun_out/r3v_split_pipe.txt; tail -3 gpurun_out/r3v_split_pipe.err
