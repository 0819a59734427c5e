cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "kn2 or golden or CRN or lms or frontend or stft" > $O/run14_tests.log 2>&1; echo "rc=$?" >> $O/run14_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run14_bench_bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/run14_counters.txt 2>&1
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_a -o a -- $B > $O/run14_pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc_b -o b -- $B > $O/run14_pmc_b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_c -o c -- $B > $O/run14_pmc_c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_d -o d -- $B > $O/run14_pmc_d.log 2>&1
cd $GRAFT_REPO_ROOT
# keep the merge under 64 MiB: counter CSVs only
find $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d -name "*kernel_trace*" -delete 2>/dev/null
du -sh $O | tail -1
tail -3 $O/run14_tests.log; tail -1 $O/run14_bench_bf16.log | cut -c1-300
