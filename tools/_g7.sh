cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python tools/optable.py --reps 5 --json $O/g7_optable.json > $O/g7_optable.log 2>&1; tail -5 $O/g7_optable.log
timeout 1200 python tools/heldout_eval.py train --steps 1000 --batch 8 --train-len 16000 --heldout 48 --pool 96 --init formula --score --no-wav \
  --legs fp32:9,bf16:9,fp32:10,bf16:10,fp32:11,bf16:11,fp32:12,bf16:12 --out $O/heldout_r04b > $O/g7_heldout.log 2>&1; tail -3 $O/g7_heldout.log
cp $O/heldout_r04b/train_log.json $O/heldout_r04/train_log_b.json; rm -rf $O/heldout_r04b
