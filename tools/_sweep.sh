cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/sw_$tag.log 2>&1; echo "$tag $(tail -1 $O/sw_$tag.log | grep -o '"ms_per_step": [0-9.]*')"; }
run base A=1
run nt SEFD_BN_NT=1
run base2 A=1
run nt2 SEFD_BN_NT=1
run base3 A=1
run nt3 SEFD_BN_NT=1
