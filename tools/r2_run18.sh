cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "FullSubNet" > $O/r2_run18_ops.log 2>&1; echo "rc=$?" >> $O/r2_run18_ops.log
tail -12 $O/r2_run18_ops.log
