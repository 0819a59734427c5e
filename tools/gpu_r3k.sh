cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "DCCRN and (E-kn0 or 2403)" > $O/r3k_tests.log 2>&1; tail -3 $O/r3k_tests.log
timeout 300 python tools/optable.py --reps 10 > $O/r3k_optable.log 2>&1; grep -E "STFT|ISTFT" $O/r3k_optable.log
