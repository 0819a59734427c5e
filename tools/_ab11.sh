cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2 3; do
run fsn --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_FB_LANE=1 run fsn_fblane --model fullsubnet --steps 20 --warmup 5
done
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp_smoke.py -x -q -m gpu -k "fsn or FullSubNet or subband or ddp or bucket" 2>&1 | tail -3
