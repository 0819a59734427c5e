cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "2401" > $O/r2_run17_ops.log 2>&1; echo "rc=$?" >> $O/r2_run17_ops.log
tail -5 $O/r2_run17_ops.log
timeout 400 python tools/opbench.py --ab "SEFD_WG256=0" "SEFD_WG256=1" > $O/r2_opbench17.log 2>&1
grep -E "WGRAD.*N  128|totals|step ms" $O/r2_opbench17.log | head
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --model dccrn_large --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
