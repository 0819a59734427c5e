cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "short_row or fused_losses" > $O/g1_new.log 2>&1; tail -3 $O/g1_new.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fullsubnet_step_against or status_word or schedule or large_at_bench" > $O/g1_new2.log 2>&1; tail -5 $O/g1_new2.log
timeout 900 python -m pytest tests/test_gpu_ddp_smoke.py -x -q -m gpu > $O/g1_ddp.log 2>&1; tail -5 $O/g1_ddp.log
timeout 600 python bench.py > $O/g1_bench.log 2>&1; tail -1 $O/g1_bench.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g1_bench_driver.log 2>&1; tail -1 $O/g1_bench_driver.log | cut -c1-200
timeout 2400 python -m pytest tests -x -q -m gpu > $O/g1_all.log 2>&1; tail -4 $O/g1_all.log
