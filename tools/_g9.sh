cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and DCCRN and 240" > $O/g9_ops.log 2>&1; tail -2 $O/g9_ops.log
for v in 0 64 0 64; do
SEFD_CG256_DBG=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g9_bench_$v.log 2>&1; echo "dbg=$v $(tail -1 $O/g9_bench_$v.log | cut -c60-160)"
done
timeout 300 python tools/optable.py --reps 5 --json $O/g9_optable.json > $O/g9_optable.log 2>&1; tail -2 $O/g9_optable.log
