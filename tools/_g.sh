cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2; do
for n in 0 2 4 6 8 12 99; do
echo -n "HOLD_SKIP=$n " ; SEFD_HOLD_SKIP=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done; done 2>&1 | tee $O/r5m_holdskip.log
