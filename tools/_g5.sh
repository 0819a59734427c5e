cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/g5_model.log 2>&1; tail -4 $O/g5_model.log
timeout 600 python -m pytest tests/test_gpu_ddp_smoke.py tests/test_gpu_validate.py -x -q -m gpu > $O/g5_ddp.log 2>&1; tail -3 $O/g5_ddp.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g5_bench.log 2>&1; tail -1 $O/g5_bench.log | cut -c60-160
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g5_bench2.log 2>&1; tail -1 $O/g5_bench2.log | cut -c60-160
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g5_prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/g5_prof.log 2>&1 )
rm -f $O/g5_prof/k_kernel_trace.csv
grep -E "finalize" $O/g5_prof/k_kernel_stats.csv | cut -c1-170
timeout 1500 python tools/heldout_eval.py train --steps 1000 --batch 8 --train-len 16000 --heldout 48 --pool 96 --init formula --score --no-wav --keep-wav 3 \
  --legs fp32:3,bf16:3,fp32:4,bf16:4,fp32:5,bf16:5,fp32:6,bf16:6,fp32:7,bf16:7,fp32:8,bf16:8 --out $O/heldout_r04 > $O/g5_heldout.log 2>&1; tail -14 $O/g5_heldout.log
rm -f $O/heldout_r04/noisy.npy
