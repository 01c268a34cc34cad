cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 1,70,85 --act 3 --reps 30 --rounds 2
$K conv 16 90 160 5 1 256 128,128 --impls 70,85 --act 3 --reps 30 --rounds 2
$K conv 16 90 160 3 3 256 128 --impls 1,70,85 --act 1 --reps 30 --rounds 2
$K conv 1 180 320 3 3 128 128,128,8 --impls 70,85 --act 2 --reps 50 --rounds 2
