cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for D in 0 1 2 4 3 7; do
echo "== dbg $D"; SEFD_WC_DBG=$D timeout 300 python tools/optable.py --reps 2 2>&1 | grep -E "RUNGEMM" | grep -E "tag 40[45] M" | grep -E "K +(768|384) "
done
