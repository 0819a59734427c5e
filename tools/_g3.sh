cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and DCCRN and not 7000 and not 1600" > $O/g3_ops.log 2>&1; tail -5 $O/g3_ops.log
for v in 1 0 1 0; do
SEFD_BN_FUSE_APPLY=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g3_bench_$v.log 2>&1; echo "fuse=$v $(tail -1 $O/g3_bench_$v.log | cut -c60-160)"
done
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/g3_model.log 2>&1; tail -5 $O/g3_model.log
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g3_prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/g3_prof.log 2>&1 )
python tools/timeline.py $O/g3_prof/k_kernel_trace.csv 1 v > $O/g3_timeline.txt 2>&1
rm -f $O/g3_prof/k_kernel_trace.csv
head -30 $O/g3_prof/k_kernel_stats.csv | cut -c1-150
