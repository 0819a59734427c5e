cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for ko in 0 1; do
  echo "KORDER=$ko" >> $O/run16_bench.log
  SEFD_RG_KORDER=$ko timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline >> $O/run16_bench.log 2>&1
done
grep -E "KORDER|ms_per_step" $O/run16_bench.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "kn2" > $O/run16_tests.log 2>&1; echo "rc=$?" >> $O/run16_tests.log
tail -3 $O/run16_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1_bf16_v7 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/run16_prof.log 2>&1
