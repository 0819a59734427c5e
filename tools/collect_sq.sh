#!/bin/bash
# SQ / TCP / TCC counters per kernel of the default bench step, single-lane schedule (SEFD_TUNING=NO_OVERLAP=1: one kernel in flight, so the counters
# of a dispatch are its own), four separate --pmc passes (kernel trace only, as gpurun requires) -> gpurun_out/sq_counters.json (+ .txt summary).
#   gpurun -- 'bash tools/collect_sq.sh'   then   cp gpurun_out/sq_counters.json profiles/rNN_sq_counters_single_lane.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra"
pass() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && SEFD_TUNING=NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $* --output-format csv -d $O/sq_$tag -o p -- $B > $O/sq_$tag.log 2>&1 ); find $O/sq_$tag -name "*kernel_trace*" -delete 2>/dev/null; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass d TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python tools/pmc_sq.py $O/sq_counters.json $(find $O/sq_a $O/sq_b $O/sq_c $O/sq_d -name "*counter_collection.csv") > $O/sq_counters.txt 2>&1; head -c 2500 $O/sq_counters.txt
find $O/sq_a $O/sq_b $O/sq_c $O/sq_d -name "*counter_collection.csv" -delete
