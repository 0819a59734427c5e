cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for pr in 0 1 0 1; do
  echo "PRIO=$pr" >> $O/run22_bench.log
  SEFD_RG_PRIO=$pr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O/run22_bench.log 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/run22_bench.log'):
    if l.startswith('PRIO'): print(l.strip())
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['roofline']['kernels']['rungemm_bf16'])
PY
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "kn2" > $O/run22_tests.log 2>&1; echo "rc=$?" >> $O/run22_tests.log
tail -3 $O/run22_tests.log
