cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
SEFD_STFT_PAIR=1 timeout 600 python -m pytest tests/test_gpu_model.py -q -k "test_module_step_against_reference_golden" > $O/r5n_pair1.log 2>&1; tail -4 $O/r5n_pair1.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "test_module_step_against_reference_golden" > $O/r5n_pair0.log 2>&1; tail -4 $O/r5n_pair0.log | cut -c1-200
