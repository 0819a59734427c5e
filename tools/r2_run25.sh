cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "512 or 1024 or (FullSubNet and bf16)" > $O/r2_run25_ops.log 2>&1; echo "rc=$?" >> $O/r2_run25_ops.log
tail -6 $O/r2_run25_ops.log
timeout 300 python bench.py --model dccrn_large --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_run25_large.log 2>&1
tail -1 $O/r2_run25_large.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernels']['lstm_gate_gemm_bf16'])"
