cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for v in 1 0; do
export SEFD_CG256_NB=$v
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra --model dccrn_large 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/large nb=$v /"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra --model fullsubnet 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/fsn nb=$v /"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra --batch 64 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b64 nb=$v /"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/default nb=$v /"
done
unset SEFD_CG256_NB
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
