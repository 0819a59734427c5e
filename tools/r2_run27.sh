cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
SEFD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2_run27_ddp.log 2>&1; echo "rc=$?" >> $O/r2_run27_ddp.log
tail -4 $O/r2_run27_ddp.log | cut -c1-700
