#!/bin/bash
# Interleaved A/B of two library builds through tools/kbench on one box: build/<old>/libpropainter_hip.so (LD_LIBRARY_PATH wins over the
# RUNPATH) against the in-tree library.  usage: gpu_ab_lib.sh <old-variant-name> <tag>
cd $GRAFT_REPO_ROOT
K=build/kbench
OLD=build/$1
run() {
  for r in 1 2; do
    echo "== old"; LD_LIBRARY_PATH=$OLD:$LD_LIBRARY_PATH $K "$@" | grep impl
    echo "== new"; $K "$@" | grep impl
  done
}
{
echo "# 3x3 K2304 cout256"; run conv 16 90 160 3 3 256 256 --impls 70 --act 1 --reps 40
echo "# 1x5 cout256"; run conv 16 90 160 1 5 256 128,128 --impls 70 --act 1 --reps 40
echo "# 5x1 cout128 +h"; run conv 16 90 160 5 1 128 128,128 --impls 70 --act 4 --late h --reps 40
echo "# 1x5 cout256 +zr"; run conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late zr --reps 40
echo "# 3x3 K1152 cout128"; run conv 16 90 160 3 3 128 128 --impls 70 --act 1 --reps 40
echo "# N1 180x320 3x3 2src +res"; run conv 1 180 320 3 3 128 128,128 --impls 70 --act 0 --res --reps 60
echo "# 180x320 cout432"; run conv 1 180 320 3 3 432 128 --impls 70 --act 0 --reps 60
echo "# 360x640 c64"; run conv 4 360 640 3 3 64 64 --impls 70 --act 1 --reps 40
echo "# 3 sources"; run conv 1 180 320 3 3 128 128,128,64 --impls 70 --act 1 --reps 60
} > gpurun_out/$2.txt 2>&1
cat gpurun_out/$2.txt
