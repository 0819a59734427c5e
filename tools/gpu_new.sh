cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -k "real_E or crn_direct or syncbn or validation or crn_module or wide_C" > $O/new_tests.log 2>&1; echo "rc=$?" >> $O/new_tests.log
tail -5 $O/new_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/new_bench.log 2>&1; tail -1 $O/new_bench.log | cut -c1-250
python - <<'PY'
import json
for l in open('gpurun_out/new_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['traffic'], j['roofline']['frac'])
PY
