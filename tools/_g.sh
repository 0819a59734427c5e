cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2 3; do
echo -n "base " ; SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/base.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
echo -n "nop  " ; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done 2>&1 | tee $O/r5q_nop.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -2
