cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python tools/opbench.py --minn 256 --tags 103 104 105 400 401 402 --ab "SEFD_CG256=0" "SEFD_CG256=1" "SEFD_CG256=1 SEFD_CG256_VAR=1" "SEFD_CG256=1 SEFD_CG256_VAR=4" "SEFD_CG256=1 SEFD_CG256_VAR=4 SEFD_CG256_DBG=8" "SEFD_CG256=1 SEFD_CG256_VAR=4 SEFD_CG256_DBG=1" "SEFD_CG256=1 SEFD_CG256_VAR=4 SEFD_CG256_DBG=2" "SEFD_CG256=1 SEFD_CG256_VAR=4 SEFD_CG256_DBG=4" "SEFD_CG256=1 SEFD_CG256_VAR=4 SEFD_CG256_DBG=15" > $O/r2_run4_opbench.log 2>&1
head -1 $O/r2_run4_opbench.log; tail -9 $O/r2_run4_opbench.log
SEFD_CG256=1 SEFD_CG256_VAR=4 timeout 600 python -m pytest tests -q -m gpu -x -k "every_op and kn3" > $O/r2_run4_ops_ws.log 2>&1; tail -3 $O/r2_run4_ops_ws.log
timeout 300 python tools/diag_lstm_rows.py 8000 > $O/r2_run4_lstm.log 2>&1; head -4 $O/r2_run4_lstm.log
timeout 900 python -m pytest tests/test_gpu_validate.py tests/test_gpu_model.py -q -m gpu -k "bf16 or validate or checkpoint or interface or istft" > $O/r2_run4_tests.log 2>&1; tail -8 $O/r2_run4_tests.log
