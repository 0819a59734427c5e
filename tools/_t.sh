cd /root/repo
python -m pytest tests/test_gpu_ops.py -q -x -k "DCCRN-1-2400-C-kn3 or DCCRN-3-4000-R-kn2" 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py -q -k "two_stream or fused_train_step or bf16_dccrn" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('on', json.loads(sys.stdin.read())['ms_per_step'])"; SEFD_TUNING=ENC0_DIRECT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('off', json.loads(sys.stdin.read())['ms_per_step'])"; done
