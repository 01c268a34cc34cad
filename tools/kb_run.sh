cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 16 90 160 1 5 256 128,128 --impls 70,108 --act 1 --late pre --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70,108 --act 3 --late pre --reps 40 --rounds 2
$K conv 16 90 160 1 5 256 128,128 --impls 70,108 --act 3 --late zr --reps 40 --rounds 2
$K conv 16 90 160 5 1 128 128,128 --impls 70,108 --act 4 --late pre --reps 40 --rounds 2
$K conv 16 90 160 5 1 128 128,128 --impls 70,108 --act 4 --late h --reps 40 --rounds 2
$K conv 16 90 160 3 3 128 128 --impls 70,108 --act 1 --late pre --reps 40 --rounds 2
