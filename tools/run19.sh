cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run19_bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/run19_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['roofline']['kernels'])
PY
timeout 1200 python -m pytest tests -x -q -m gpu -k "kn2 or kn0 or golden or CRN or FullSubNet" > $O/run19_tests.log 2>&1; echo "rc=$?" >> $O/run19_tests.log
tail -3 $O/run19_tests.log
