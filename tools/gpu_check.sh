# usage (on the GPU box): bash tools/gpu_check.sh <tag> "<pytest -k expression>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
TAG=$1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/${TAG}_bench.log'):
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['value'], j['final_loss'], j['roofline']['kernels'])
PY
timeout 1500 python -m pytest tests -x -q -m gpu -k "$2" > $O/${TAG}_tests.log 2>&1; echo "rc=$?" >> $O/${TAG}_tests.log
tail -3 $O/${TAG}_tests.log
