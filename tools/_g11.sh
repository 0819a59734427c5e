cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4 > $O/g11_tests.log; cat $O/g11_tests.log
for i in 1 2 3; do
for v in new base; do
if [ $v = base ]; then export SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/base.so; else unset SEFD_LIB_PATH; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g11_$v$i.log 2>&1; echo "$v $(tail -1 $O/g11_$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
unset SEFD_LIB_PATH
timeout 900 python tools/opbench.py --minn 64 --ab "SEFD_X=0" "SEFD_X=1" "SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/base.so" > $O/g11_opbench.log 2>&1; tail -75 $O/g11_opbench.log | cut -c1-200
