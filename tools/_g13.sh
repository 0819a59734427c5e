cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
SEFD_CG256_NB=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bf16" 2>&1 | tail -3 > $O/g13_tests.log; cat $O/g13_tests.log
for i in 1 2 3; do
for v in 1 0; do
SEFD_CG256_NB=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g13_nb$v$i.log 2>&1; echo "nb=$v $(tail -1 $O/g13_nb$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
timeout 900 python tools/opbench.py --minn 256 --ab "SEFD_CG256_NB=0" "SEFD_CG256_NB=1" "SEFD_CG256_NB=1 SEFD_CG256_DBG=8" "SEFD_CG256_NB=1 SEFD_CG256_DBG=2" "SEFD_CG256_NB=1 SEFD_CG256_DBG=1" > $O/g13_opbench.log 2>&1; grep -E "GEMM |totals|step ms" $O/g13_opbench.log | cut -c1-230
