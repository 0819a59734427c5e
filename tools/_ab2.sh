cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run default --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=64,MAIN_CUS=192 run s64_m192 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=96,MAIN_CUS=160 run s96_m160 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=128,MAIN_CUS=128 run s128_m128 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=32,MAIN_CUS=224 run s32_m224 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=64,MAIN_CUS=256 run s64_m256 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=128,MAIN_CUS=256 run s128_m256 --steps 20 --warmup 5
SEFD_TUNING=MAIN_CUS=256 run sfree_m256 --steps 20 --warmup 5
SEFD_TUNING=SIDE_CUS=255,MAIN_CUS=256 run s255_m256 --steps 20 --warmup 5
done
