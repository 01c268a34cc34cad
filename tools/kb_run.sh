cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 70,109 --act 3 --late zr --reps 40 --rounds 3
$K conv 16 90 160 5 1 128 128,128 --impls 70,109 --act 4 --late h --reps 40 --rounds 3
$K conv 16 90 160 1 5 256 128,128 --impls 70,109 --act 1 --reps 40 --rounds 3
$K conv 16 90 160 3 3 128 128 --impls 70,109 --act 1 --reps 40 --rounds 3
$K conv 1 180 320 3 3 128 128,128 --impls 70,109 --act 0 --res --reps 60 --rounds 3
