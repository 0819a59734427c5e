cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "2401 or (DCCRN and 2400 and bf16)" > $O/r2_run16_ops.log 2>&1; echo "rc=$?" >> $O/r2_run16_ops.log
tail -12 $O/r2_run16_ops.log
timeout 400 python tools/opbench.py --ab "SEFD_WG256=0" "SEFD_WG256=1" > $O/r2_opbench16.log 2>&1
grep -E "WGRAD|totals|step ms" $O/r2_opbench16.log | head -80
for s in 2 3; do SEFD_WG256_STAGES=$s timeout 300 python tools/opbench.py --minn 256 2>&1 | grep -E "WGRAD.*N  256|totals" | sed "s/^/S$s /" ; done > $O/r2_opbench16_stages.log 2>&1
cat $O/r2_opbench16_stages.log | head -40
