cd $GRAFT_REPO_ROOT
K=build/kbench
$K conv 1 1 109140 1 1 1960 512 --impls 80,12,13,15,17 --reps 30 --rounds 2
$K conv 1 1 109140 1 1 1536 512 --impls 80,12,13 --reps 30 --rounds 2
$K conv 1 1 109140 1 1 512 512 --impls 80,12,13 --res --reps 30 --rounds 2
$K conv 1 1 109140 1 1 6272 512 --impls 80,12,13 --reps 20 --rounds 2
