cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "DCCRN" > $O/r2_run28_ops.log 2>&1; echo "rc=$?" >> $O/r2_run28_ops.log
tail -4 $O/r2_run28_ops.log
timeout 400 python tools/opbench.py --ab "SEFD_WG_DUAL=0" "SEFD_WG_DUAL=1" > $O/r2_opbench28.log 2>&1
grep -E "WGRAD.*N  (128|256) K [ 0-9]{4} |totals|step ms" $O/r2_opbench28.log | head -20
for v in 0 1; do SEFD_WG_DUAL=$v timeout 300 python bench.py --model dccrn_large --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c50-140; done
for v in 0 1; do SEFD_WG_DUAL=$v timeout 300 python bench.py --model fullsubnet --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c50-140; done
