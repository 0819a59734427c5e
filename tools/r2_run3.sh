cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python tools/opbench.py --minn 128 --tags 102 103 104 105 400 401 402 403 --ab "SEFD_CG256=0" "SEFD_CG256=3" "SEFD_CG256=3 SEFD_CG256_VAR=1" "SEFD_CG256=3 SEFD_CG256_VAR=2" "SEFD_CG256=3 SEFD_CG256_DBG=16" "SEFD_CG256=3 SEFD_CG256_DBG=8" "SEFD_CG256=3 SEFD_CG256_DBG=1" "SEFD_CG256=3 SEFD_CG256_DBG=2" "SEFD_CG256=3 SEFD_CG256_DBG=4" "SEFD_CG256=3 SEFD_CG256_DBG=32" "SEFD_CG256=3 SEFD_CG256_DBG=3" "SEFD_CG256=3 SEFD_CG256_DBG=15" > $O/r2_run3_opbench.log 2>&1
head -1 $O/r2_run3_opbench.log; tail -12 $O/r2_run3_opbench.log
for e in "" "SEFD_LSTM_CHUNKS=1" "SEFD_NO_OVERLAP=1"; do env $e timeout 300 python tools/diag_lstm_rows.py 8000 > "$O/r2_run3_lstm_${e}.log" 2>&1; done
tail -n 20 $O/r2_run3_lstm_*.log
