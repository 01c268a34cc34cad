cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late pre --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late pre --late16 --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late zr --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --late zr --late16 --reps 40 --rounds 2
$K conv 16 90 160 1 5 128 128,128 --impls 70 --act 4 --late h --reps 40 --rounds 2
$K conv 16 90 160 1 5 128 128,128 --impls 70 --act 4 --late h --late16 --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70 --act 3 --reps 40 --rounds 2
