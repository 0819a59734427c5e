cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "bf16 and (default or 2401 or large or schedule)" 2>&1 | tail -2
for i in 1 2 3; do
for v in 1 0; do
SEFD_WG_SPREAD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g19_s$v$i.log 2>&1; echo "spread=$v $(tail -1 $O/g19_s$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
timeout 900 python tools/opbench.py --minn 256 --ab "SEFD_WG_SPREAD=0" "SEFD_WG_SPREAD=1" > $O/g19_opbench.log 2>&1; grep -E "WGRAD|totals|step ms" $O/g19_opbench.log | cut -c1-170 | head -30
