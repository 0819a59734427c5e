cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2; do
for n in 0 1; do
echo -n "NO_OPHOLD=$n " ; SEFD_NO_OPHOLD=$n timeout 600 python bench.py --model fullsubnet --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done; done 2>&1 | tee $O/r5o_fsn_hold.log
