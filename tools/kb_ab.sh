#!/bin/bash
# In-call A/B of two library builds (tools/build_variant.sh) with tools/kbench: tools/kb_ab.sh <libA> <libB>  (run on the GPU box)
cd $GRAFT_REPO_ROOT
K=build/kbench
A=build/$1; B=build/$2
run() {
  for rep in 1 2; do
    for L in $A $B; do
      echo "== $L : $*"
      LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH $K "$@" 2>&1 | grep -v "^$" | tail -4
    done
  done
}
run conv 16 90 160 5 1 128 128,128 --impls 70 --act 4 --late h --reps 40 --rounds 2
run conv 16 90 160 5 1 256 128,128 --impls 70 --act 3 --late zr --reps 40 --rounds 2
run conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late zr --reps 40 --rounds 2
run conv 16 90 160 3 3 128 128 --impls 70 --act 1 --reps 40 --rounds 2
run conv 16 90 160 3 3 256 256 --impls 70 --act 1 --reps 40 --rounds 2
run conv 1 180 320 3 3 128 128,128 --impls 70 --act 0 --res --reps 60 --rounds 2
run conv 2 360 640 3 3 128 128 --impls 70 --act 1 --reps 30 --rounds 2
