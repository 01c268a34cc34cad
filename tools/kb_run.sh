cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 3 3 192 256 --impls 70,71,72,12,13 --act 1 --reps 30 --rounds 2
$K conv 16 90 160 3 3 126 192,64 --impls 70,71,72,12 --act 1 --reps 30 --rounds 2
$K conv 16 90 160 1 1 256 328 --impls 0,12,13,22 --act 1 --reps 30 --rounds 2
$K conv 1 180 320 3 3 128 128,128,8 --impls 0,12,13,22 --act 2 --reps 50 --rounds 2
$K conv 1 180 320 3 3 432 128 --impls 70,71,72,12 --reps 50 --rounds 2
$K conv 2 360 640 3 3 64 64 --impls 0,70,72,12,22 --act 2 --reps 30 --rounds 2
