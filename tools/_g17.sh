cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1700 python tools/heldout_eval.py train --steps 1000 --batch 8 --train-len 16000 --heldout 48 --pool 96 --init formula --score --no-wav \
  --legs fp32:13,bf16:13,fp32:14,bf16:14,fp32:15,bf16:15,fp32:16,bf16:16,fp32:17,bf16:17,fp32:18,bf16:18,fp32:19,bf16:19,fp32:20,bf16:20 --out $O/heldout_r04c > $O/g17_heldout.log 2>&1; tail -2 $O/g17_heldout.log
rm -f $O/heldout_r04c/*.npy
