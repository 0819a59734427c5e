cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for bm in 256 257; do
  echo "BM=$bm" >> $O/run17_bench.log
  SEFD_RG_BM=$bm timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O/run17_bench.log 2>&1
done
grep -E "BM=|ms_per_step" $O/run17_bench.log | cut -c1-220
python - <<'PY'
import json
for l in open('gpurun_out/run17_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['kernels'])
PY
SEFD_RG_BM=256 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "default or full" > $O/run17_tests.log 2>&1; echo "rc=$?" >> $O/run17_tests.log
tail -3 $O/run17_tests.log
