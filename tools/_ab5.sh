cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and FullSubNet" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fsn or FullSubNet or fullsubnet or subband" 2>&1 | tail -5
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run cat2 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WGCAT2=0 run separate --model fullsubnet --steps 20 --warmup 5
done
