cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and bf16" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bf16" 2>&1 | tail -5
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2 3; do
run fused --steps 20 --warmup 5
SEFD_TUNING=ENC0_BNFUSE=0 run unfused --steps 20 --warmup 5
done
