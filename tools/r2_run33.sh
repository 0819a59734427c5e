cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/opbench.py --ab "SEFD_WG_SKIP=0" "SEFD_WG_SKIP=1" > $O/r2_opbench33.log 2>&1
grep -E "WGRAD|totals|step ms" $O/r2_opbench33.log | head -50
