cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for S in 3 4; do
SEFD_RG_STAGES=$S timeout 300 python tools/optable.py --json $O/r3c_optable_s$S.json > $O/r3c_optable_s$S.log 2>&1; tail -3 $O/r3c_optable_s$S.log
done
