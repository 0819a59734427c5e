cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export SEFD_CG256_NB=1
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bf16" 2>&1 | tail -3 > $O/g14_tests.log; cat $O/g14_tests.log
for i in 1 2 3; do
for v in new prev; do
if [ $v = prev ]; then export SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/prev.so; else unset SEFD_LIB_PATH; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g14_$v$i.log 2>&1; echo "$v $(tail -1 $O/g14_$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
unset SEFD_LIB_PATH
timeout 900 python tools/opbench.py --minn 256 --ab "SEFD_X=0" "SEFD_LIB_PATH=$GRAFT_REPO_ROOT/ab/prev.so" "SEFD_CG256_DBG=8" "SEFD_CG256_DBG=2" "SEFD_CG256_DBG=1" > $O/g14_opbench.log 2>&1; grep -E "GEMM |totals|step ms" $O/g14_opbench.log | cut -c1-230 | grep -v "tag 201"
