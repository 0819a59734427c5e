cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g6_bench$i.log 2>&1; tail -1 $O/g6_bench$i.log | cut -c60-160; done
timeout 2700 python -m pytest tests -q -m gpu > $O/g6_all.log 2>&1; tail -8 $O/g6_all.log
