cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2 3; do
run default --steps 20 --warmup 5
SEFD_CG128=0 run cg128off --steps 20 --warmup 5
done
