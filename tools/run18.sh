cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for dbg in 0 32 64 96; do
  echo "DBG=$dbg" >> $O/run18_bench.log
  SEFD_RG_DBG=$dbg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline >> $O/run18_bench.log 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/run18_bench.log'):
    if l.startswith('DBG'): print(l.strip())
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['kernels'])
PY
