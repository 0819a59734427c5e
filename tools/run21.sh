cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run21_bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/run21_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['final_loss'])
PY
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "kn2 or kn0 or CRN" > $O/run21_tests.log 2>&1; echo "rc=$?" >> $O/run21_tests.log
tail -3 $O/run21_tests.log
cd /tmp && export TMPDIR=/tmp
SEFD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1_bf16_v8 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/run21_prof.log 2>&1
