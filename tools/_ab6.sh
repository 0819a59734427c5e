cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
tl() { local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_prof_$tag -o k -- python $GRAFT_REPO_ROOT/bench.py $* --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $O/x_prof_$tag.log 2>&1 ) ; }
tl default
python tools/timeline.py $O/x_prof_default/k_kernel_trace.csv 1 v > $O/x_timeline_default.txt 2>&1
tl fullsubnet --model fullsubnet
TIMELINE_MARK=fsn_in_kernel:1 python tools/timeline.py $O/x_prof_fullsubnet/k_kernel_trace.csv 1 v > $O/x_timeline_fullsubnet.txt 2>&1
find $O -name "k_kernel_trace.csv" -delete 2>/dev/null
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
for i in 1 2; do
run default --steps 20 --warmup 5
SEFD_TUNING=BN_FUSE=2 run bnfuse2 --steps 20 --warmup 5
done
for i in 1 2; do
run fsn_r8 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=4 run fsn_r4 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=6 run fsn_r6 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=12 run fsn_r12 --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_WG_ROUNDS=16 run fsn_r16 --model fullsubnet --steps 20 --warmup 5
done
