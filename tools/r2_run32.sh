cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "7000 or 2401 or FullSubNet" > $O/r2_run32_ops.log 2>&1; tail -3 $O/r2_run32_ops.log
timeout 400 python tools/opbench.py > $O/r2_opbench32.log 2>&1
grep -E "WGRAD.*N  (128|256) K [ 0-9]{4} |totals|step ms" $O/r2_opbench32.log | head -20
for m in dccrn dccrn_large fullsubnet; do timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c50-150; done
