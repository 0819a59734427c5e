cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "7000 or FullSubNet" > $O/r2_run34_ops.log 2>&1; tail -3 $O/r2_run34_ops.log
timeout 600 python tools/opbench.py --ab "SEFD_WG_SKIP=0" "SEFD_WG_SKIP=1" > $O/r2_opbench34.log 2>&1
grep -E "WGRAD.*N  256|totals|step ms" $O/r2_opbench34.log | head -20
for v in 0 1 0 1; do SEFD_WG_SKIP=$v timeout 300 python bench.py --model fullsubnet --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c50-140; done
