cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x -k "every_op and 2401" > $O/r2_run5_ops.log 2>&1; tail -5 $O/r2_run5_ops.log
timeout 1500 python tools/opbench.py --minn 256 --tags 103 104 105 400 401 402 --ab "SEFD_CG256=0" "SEFD_CG256=1" "SEFD_CG256=1 SEFD_CG256_DBG=8" "SEFD_CG256=1 SEFD_CG256_DBG=1" "SEFD_CG256=1 SEFD_CG256_DBG=2" "SEFD_CG256=1 SEFD_CG256_DBG=4" "SEFD_CG256=1 SEFD_CG256_DBG=32" "SEFD_CG256=1 SEFD_CG256_DBG=15" > $O/r2_run5_opbench.log 2>&1
head -1 $O/r2_run5_opbench.log; tail -8 $O/r2_run5_opbench.log
