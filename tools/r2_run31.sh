cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "7000 or 2401" > $O/r2_run31_ops.log 2>&1; tail -3 $O/r2_run31_ops.log
for v in 0 1 0 1; do SEFD_BN_FIN2=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c50-150; done
