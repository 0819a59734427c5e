cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bf16" 2>&1 | tail -1
for i in 1 2 3; do
for v in new prev; do
if [ $v = prev ]; then export SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/prev.so; else unset SEFD_LIB_PATH; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g18_$v$i.log 2>&1; echo "$v $(tail -1 $O/g18_$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
