cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and bf16 and (2400 or 11)" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "reproduc or schedule or golden" 2>&1 | tail -3
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run default --steps 20 --warmup 5
run fsn --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=RG_STAGES=3 run fsn_rg3 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=RG_STAGES=4 run fsn_rg4 --model fullsubnet --steps 20 --warmup 5
run large --model dccrn_large --steps 20 --warmup 5
done
