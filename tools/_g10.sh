cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "schedule or golden" 2>&1 | tail -3 > $O/g10_tests.log; cat $O/g10_tests.log
for i in 1 2 3; do
for v in 1 0; do
SEFD_LSTM_LANE3=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g10_l$v$i.log 2>&1; echo "lane3=$v $(tail -1 $O/g10_l$v$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
for c in 2 3 6 8; do
SEFD_LSTM_CHUNKS=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/g10_c$c.log 2>&1; echo "chunks=$c $(tail -1 $O/g10_c$c.log | grep -o '"ms_per_step": [0-9.]*')"
done
timeout 900 python tools/opbench.py --minn 256 --ab "SEFD_CG256_DBG=0" "SEFD_CG256_DBG=32" "SEFD_CG256_DBG=2" "SEFD_CG256_DBG=1" "SEFD_CG256_DBG=4" "SEFD_CG256_DBG=8" > $O/g10_opbench.log 2>&1; tail -40 $O/g10_opbench.log | cut -c1-260
