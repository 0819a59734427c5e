cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for v in 0 1 0 1; do
  SEFD_BN_REV=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > $O/r2_run23_$v.log 2>&1
  echo "BN_REV=$v $(tail -1 $O/r2_run23_$v.log | cut -c50-150)"
done
timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "default_E or small_C" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
SEFD_BN_REV=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bnrev -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run23_prof.log 2>&1
find $O/prof_bnrev -name "*kernel_trace*" -delete
grep "bn_" $O/prof_bnrev/r2_kernel_stats.csv | cut -c1-150
