cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for v in 0 34; do
  SEFD_ROWS_FWD=$v timeout 300 python bench.py --model fullsubnet --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run22_$v.log 2>&1
  echo "FWD=$v $(tail -1 $O/r2_run22_$v.log | cut -c50-150)"
  SEFD_ROWS_FWD=$v python tools/opbench_fsn.py 2>&1 | grep -E "LSTM_FWD.*1536"
done
