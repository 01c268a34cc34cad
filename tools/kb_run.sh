cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 70,82,12 --act 1 --reps 30 --rounds 3
$K conv 16 90 160 3 3 128 128 --impls 70,82,12 --act 1 --reps 30 --rounds 3
$K conv 16 90 160 3 3 192 256 --impls 70,82 --act 1 --reps 30 --rounds 3
$K conv 16 90 160 5 1 128 128,128 --impls 70,82 --act 1 --reps 30 --rounds 3
$K conv 1 180 320 3 3 128 128 --impls 70,82 --act 2 --reps 50 --rounds 3
