cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_op and FullSubNet" 2>&1 | tail -3
grep -n "tag  203" gpurun_out/ops_report_FullSubNet_B1_E_bf16_11.txt | head -8
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fsn or FullSubNet or subband" 2>&1 | tail -3
run() { tag=$1; shift; python bench.py "$@" --no-cpu-baseline --no-roofline --no-extra 2>&1 | tail -5 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d.get('final_loss'))"; }
for i in 1 2 3; do
run split --model fullsubnet --steps 20 --warmup 5
SEFD_TUNING=FSN_DXSPLIT=0 run one --model fullsubnet --steps 20 --warmup 5
done
