cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 2 90 160 3 3 128 128,128,128 --impls 71,72,12 --act 2 --reps 100 --rounds 3
$K conv 2 90 160 3 3 128 128,128 --impls 71,72,12 --act 2 --reps 100 --rounds 3
$K conv 2 90 160 3 3 128 128 --impls 71,72,12 --act 2 --reps 100 --rounds 3
$K conv 1 180 320 3 3 128 128,128 --impls 71,72,12 --act 2 --reps 100 --rounds 3
$K conv 1 180 320 3 3 128 128 --impls 71,72,12 --act 2 --reps 100 --rounds 3
