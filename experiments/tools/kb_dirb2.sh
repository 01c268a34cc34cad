#!/bin/bash
# DIRB with fragment-major weights (impl 117) against the shipped halo kernel (impl 70); PP_DIAG library under build/$1
cd $GRAFT_REPO_ROOT
K=build/kbench
export LD_LIBRARY_PATH=build/$1:$LD_LIBRARY_PATH
$K conv 16 90 160 1 5 256 128,128 --impls 70,117,70,117,118 --act 1 --reps 40 --prof
$K conv 16 90 160 5 1 256 128,128 --impls 70,117,70,117 --act 1 --reps 40
$K conv 16 90 160 5 1 128 128,128 --impls 70,117,70,117 --act 4 --late h --reps 40
$K conv 16 90 160 3 3 256 256 --impls 70,117,70,117 --act 1 --reps 40
$K conv 16 90 160 3 3 128 128 --impls 70,117,70,117 --act 1 --reps 40
