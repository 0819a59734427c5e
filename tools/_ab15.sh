cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2 3; do
run r8 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=4 run r4 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=2 run r2 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=3 run r3 --model fullsubnet --steps 20 --warmup 5
done
