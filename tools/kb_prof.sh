#!/bin/bash
# Phase stamps (impl 75) of the halo kernel through tools/kbench on a PP_DIAG build: tools/kb_prof.sh <lib>
cd $GRAFT_REPO_ROOT
K=build/kbench
export LD_LIBRARY_PATH=build/$1:$LD_LIBRARY_PATH
$K conv 16 90 160 3 3 256 256 --impls 75 --act 1 --reps 40 --prof
$K conv 16 90 160 1 5 256 128,128 --impls 75 --act 1 --reps 40 --prof
$K conv 16 90 160 3 3 128 128 --impls 75 --act 1 --reps 40 --prof
$K conv 16 90 160 3 3 64 128 --impls 75 --act 1 --reps 40 --prof
$K conv 1 180 320 3 3 128 128 --impls 75 --act 1 --reps 60 --prof
