cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and DCCRN and 2400" > $O/g4_ops.log 2>&1; tail -3 $O/g4_ops.log
for v in 1 0 1 0; do
SEFD_BN_FUSE_APPLY=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g4_bench_$v.log 2>&1; echo "fuse=$v $(tail -1 $O/g4_bench_$v.log | cut -c60-160)"
done
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g4_prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/g4_prof.log 2>&1 )
rm -f $O/g4_prof/k_kernel_trace.csv
grep -E "rungemm|wgrad|bn_apply" $O/g4_prof/k_kernel_stats.csv | cut -c1-170
