cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $O/g8_counters.txt 2>&1 )
grep -c . $O/g8_counters.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra"
pass() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && SEFD_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc $* --output-format csv -d $O/g8_$tag -o p -- $B > $O/g8_$tag.log 2>&1 ); find $O/g8_$tag -name "*kernel_trace*" -delete 2>/dev/null; ls $O/g8_$tag/*counter_collection.csv 2>/dev/null | head -1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass d TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python tools/pmc_sq.py $O/g8_sq.json $(find $O/g8_a $O/g8_b $O/g8_c $O/g8_d -name "*counter_collection.csv") > $O/g8_sq.txt 2>&1; head -c 3000 $O/g8_sq.txt
find $O/g8_a $O/g8_b $O/g8_c $O/g8_d -name "*counter_collection.csv" -delete
timeout 900 python tools/heldout_eval.py train --steps 1000 --batch 8 --train-len 16000 --heldout 48 --pool 96 --init formula --score --no-wav \
  --legs fp32:9,bf16:9,fp32:10,bf16:10,fp32:11,bf16:11,fp32:12,bf16:12 --out $O/heldout_r04b > $O/g8_heldout.log 2>&1; tail -2 $O/g8_heldout.log
rm -f $O/heldout_r04b/*.npy
