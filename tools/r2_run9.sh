# bench validation (all models), kernel-trace profile and PMC traffic passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 700 python bench.py > $O/r2_run9_bench.log 2>&1; echo "bench rc=$?" >> $O/r2_run9_bench.log
tail -2 $O/r2_run9_bench.log | cut -c1-600
for m in dccrn_large fullsubnet; do
  timeout 400 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_run9_bench_$m.log 2>&1; echo "rc=$?" >> $O/r2_run9_bench_$m.log
  tail -2 $O/r2_run9_bench_$m.log | cut -c1-400
done
timeout 400 python bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_run9_bench_b64.log 2>&1; echo "rc=$?" >> $O/r2_run9_bench_b64.log
tail -2 $O/r2_run9_bench_b64.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r2a -o r2 -- $B > $O/r2_run9_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_r2_c -o c -- $B > $O/r2_run9_pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_r2_d -o d -- $B > $O/r2_run9_pmc_d.log 2>&1
find $O/pmc_r2_c $O/pmc_r2_d $O/prof_r2a -name "*kernel_trace*" -delete 2>/dev/null
ls -la $O/prof_r2a/* $O/pmc_r2_c/* | head
