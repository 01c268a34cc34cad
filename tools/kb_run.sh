cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 40 90 160 1 5 256 128,128 --impls 70,76,83 --act 3 --late zr --reps 20
$K conv 40 90 160 5 1 128 128,128 --impls 70,76,83 --act 4 --late h --reps 20
$K conv 40 90 160 5 1 128 128,128 --impls 70,76,83 --act 4 --reps 20
