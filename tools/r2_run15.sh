cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "DCCRN and bf16 or CRN" > $O/r2_run15_ops.log 2>&1; echo "rc=$?" >> $O/r2_run15_ops.log
tail -12 $O/r2_run15_ops.log
timeout 300 python tools/opbench.py > $O/r2_opbench15.log 2>&1
grep -E "WGRAD|totals|step ms" $O/r2_opbench15.log
