cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
run() { echo "== $1"; env $1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175; }
run A=1
run SEFD_SPLITSUM_MID=0
run A=2
echo "== large"; timeout 600 python bench.py --model dccrn_large --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
echo "== fsn"; timeout 600 python bench.py --model fullsubnet --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
echo "== pmsqe"; timeout 600 python bench.py --perceptual PMSQE --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c60-175
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "DCCRN or dccrn" > $O/t_tests.log 2>&1; echo "rc=$?" >> $O/t_tests.log; tail -3 $O/t_tests.log | cut -c1-200
