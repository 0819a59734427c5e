cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/opbench.py --ab "SEFD_CG256=0" "SEFD_CG256=3" > $O/r2_run2_opbench.log 2>&1
head -3 $O/r2_run2_opbench.log; tail -3 $O/r2_run2_opbench.log
timeout 900 python -m pytest tests -q -m gpu -x -k "every_op and (kn3 or kn1 or kn7)" > $O/r2_run2_ops.log 2>&1; tail -3 $O/r2_run2_ops.log
timeout 900 python -m pytest tests/test_gpu_validate.py tests/test_gpu_model.py -q -m gpu --durations=8 > $O/r2_run2_tests.log 2>&1; tail -12 $O/r2_run2_tests.log
for a in "48000 bf16" "8000 bf16" "48000 fp32"; do timeout 300 python tools/diag_batch_indep.py $a > "$O/r2_run2_diag_${a// /_}.log" 2>&1; done
head -1 $O/r2_run2_diag_*.log
