cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and DCCRN and bf16 and (2400 or 4000)" 2>&1 | tail -3
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2 3; do
run s1024 --steps 20 --warmup 5
SEFD_TUNING=ENC0_WG_SLOTS=768 run s768 --steps 20 --warmup 5
SEFD_TUNING=ENC0_WG_SLOTS=512 run s512 --steps 20 --warmup 5
SEFD_TUNING=ENC0_WG_SLOTS=2048 run s2048 --steps 20 --warmup 5
done
python tools/opbench.py --tags 100 2>&1 | grep -i "wgrad\|enc0" | head
for i in 1 2; do
run fsn --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_HOLD=0 run fsn_nohold --model fullsubnet --steps 20 --warmup 5
done
