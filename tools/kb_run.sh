cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 70,85,86,87 --act 3 --reps 40 --rounds 3
$K conv 16 90 160 3 3 128 128 --impls 70,85,86,87 --act 1 --reps 40 --rounds 3
$K conv 16 90 160 3 3 192 256 --impls 70,85,86,87 --act 1 --reps 40 --rounds 3
