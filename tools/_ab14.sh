cd /root/repo
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$tag', d['ms_per_step'], {x:k[x] for x in k if 'stft' in x})"; }
for i in 1 2; do
run ppw1 --steps 20 --warmup 5
SEFD_TUNING=STFT_PPW=2 run ppw2 --steps 20 --warmup 5
done
