#!/bin/bash
cd $GRAFT_REPO_ROOT
K=build/kbench
export LD_LIBRARY_PATH=build/$1:$LD_LIBRARY_PATH
$K conv 16 90 160 1 5 256 128,128 --impls 70,112,75,113 --act 1 --reps 40 --rounds 2 --prof
$K conv 16 90 160 5 1 256 128,128 --impls 70,112 --act 1 --reps 40 --rounds 2
$K conv 16 90 160 5 1 128 128,128 --impls 70,112 --act 4 --late h --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70,112 --act 3 --late zr --reps 40 --rounds 2
$K conv 16 90 160 3 3 256 256 --impls 70,112 --act 1 --reps 40 --rounds 2
