cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run large --model dccrn_large --steps 20 --warmup 5
SEFD_TUNING=ENC0_BNFUSE=0 run large_nofuse --model dccrn_large --steps 20 --warmup 5
SEFD_TUNING=ENC0_DIRECT=0 run large_noenc0 --model dccrn_large --steps 20 --warmup 5
done
