cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run23_bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/run23_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['final_loss'])
PY
timeout 1200 python -m pytest tests -x -q -m gpu -k "kn2 or golden or CRN" > $O/run23_tests.log 2>&1; echo "rc=$?" >> $O/run23_tests.log
tail -3 $O/run23_tests.log
