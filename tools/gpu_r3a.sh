# round-3 first call: new tests + baseline bench + kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "bench_size or bf16_fullsubnet or pmsqe or validation_path" > $O/r3a_tests.log 2>&1; echo "rc=$?" >> $O/r3a_tests.log
tail -5 $O/r3a_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/r3a_bench.log 2>&1; tail -1 $O/r3a_bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3a_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/r3a_prof.log 2>&1
ls $O/r3a_prof | head
