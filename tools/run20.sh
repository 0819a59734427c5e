cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
echo "NO_OVERLAP" >> $O/run20_bench.log
SEFD_NO_OVERLAP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O/run20_bench.log 2>&1
echo "OVERLAP" >> $O/run20_bench.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O/run20_bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/run20_bench.log'):
    if not l.startswith('{'): print(l.strip())
    else:
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['final_loss'])
PY
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/run20_tests.log 2>&1; echo "rc=$?" >> $O/run20_tests.log
tail -3 $O/run20_tests.log
