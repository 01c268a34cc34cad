#!/bin/bash
cd $GRAFT_REPO_ROOT
K=build/kbench
export LD_LIBRARY_PATH=build/$1:$LD_LIBRARY_PATH
$K conv 16 90 160 3 3 256 256 --impls 12,70,111,75 --act 1 --reps 40 --rounds 2 --prof
$K conv 16 90 160 1 5 256 128,128 --impls 12,70,111,75 --act 1 --reps 40 --rounds 2 --prof
$K conv 16 90 160 3 3 128 128 --impls 12,70,111 --act 1 --reps 40 --rounds 2
$K conv 16 90 160 5 1 128 128,128 --impls 12,70,111 --act 4 --late h --reps 40 --rounds 2
$K conv 16 90 160 3 3 64 128 --impls 12,70 --act 1 --reps 40 --rounds 2
$K conv 16 90 160 3 3 2 128 --impls 12,70 --act 0 --reps 40 --rounds 2
$K conv 1 180 320 3 3 128 128,128 --impls 12,70,111 --act 0 --res --reps 60 --rounds 2
