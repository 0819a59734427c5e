cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run default --steps 20 --warmup 5
SEFD_TUNING=SKIP_EXP=1 run exp1_g192 --steps 20 --warmup 5
SEFD_TUNING=SKIP_EXP=2 run exp2_g192 --steps 20 --warmup 5
SEFD_TUNING=SKIP_EXP=2,SIDE_GRID=224 run exp2_g224 --steps 20 --warmup 5
SEFD_TUNING=SKIP_EXP=2,SIDE_GRID=256 run exp2_g256 --steps 20 --warmup 5
SEFD_TUNING=SKIP_EXP=2,SIDE_GRID=128 run exp2_g128 --steps 20 --warmup 5
done
