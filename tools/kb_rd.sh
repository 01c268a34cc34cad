#!/bin/bash
cd $GRAFT_REPO_ROOT
K=build/kbench
export LD_LIBRARY_PATH=build/$1:$LD_LIBRARY_PATH
A=${2:-70}; B=${3:-114}
$K conv 16 90 160 3 3 256 256 --impls $A,$B,75,115 --act 1 --reps 40 --rounds 2 --prof
$K conv 16 90 160 1 5 256 128,128 --impls $A,$B --act 1 --reps 40 --rounds 2
$K conv 16 90 160 5 1 128 128,128 --impls $A,$B --act 4 --late h --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls $A,$B --act 3 --late zr --reps 40 --rounds 2
$K conv 16 90 160 3 3 128 128 --impls $A,$B --act 1 --reps 40 --rounds 2
$K conv 1 180 320 3 3 128 128,128 --impls $A,$B --act 0 --res --reps 60 --rounds 2
$K conv 2 360 640 3 3 128 128 --impls $A,$B --act 1 --reps 30 --rounds 2
