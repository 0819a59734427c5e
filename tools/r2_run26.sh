cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for v in 0 1 0 1; do
  SEFD_LSTM_HEADFUSE=$v timeout 300 python bench.py --model fullsubnet --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run26_$v.log 2>&1
  echo "HEADFUSE=$v $(tail -1 $O/r2_run26_$v.log | cut -c50-150)"
done
