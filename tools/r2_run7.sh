cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
SEFD_CG256_VAR=1 timeout 600 python -m pytest tests -q -m gpu -x -k "every_op and 2401" > $O/r2_run7_ops.log 2>&1; tail -5 $O/r2_run7_ops.log
timeout 1500 python tools/opbench.py --minn 256 --tags 103 104 105 400 401 402 --ab "SEFD_CG256=0" "SEFD_CG256=1" "SEFD_CG256=1 SEFD_CG256_VAR=1" "SEFD_CG256=1 SEFD_CG256_VAR=1 SEFD_CG256_DBG=8" "SEFD_CG256=1 SEFD_CG256_VAR=1 SEFD_CG256_DBG=1" "SEFD_CG256=1 SEFD_CG256_VAR=1 SEFD_CG256_DBG=2" "SEFD_CG256=1 SEFD_CG256_VAR=1 SEFD_CG256_DBG=32" > $O/r2_run7_opbench.log 2>&1
head -1 $O/r2_run7_opbench.log; tail -7 $O/r2_run7_opbench.log
