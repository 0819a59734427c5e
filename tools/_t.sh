cd /root/repo
python -m pytest tests/test_gpu_model.py -q -k "bf16" 2>&1 | tail -6
python -m pytest tests/test_gpu_ops.py -q -x -k "DCCRN-3-4000-R-kn2 or DCCRN-1-2400-C-kn3 or DCCRN-2-1600" 2>&1 | tail -3
python tools/opbench.py --tags 100 --ab "" "ENC0_DIRECT=0" 2>&1 | tail -5
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('on', json.loads(sys.stdin.read())['ms_per_step'])"; SEFD_TUNING=ENC0_DIRECT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>&1 | grep ^{ | python -c "import sys,json; print('off', json.loads(sys.stdin.read())['ms_per_step'])"; done
python -c "
import json; d=json.load(open('gpurun_out/bf16_parity.json'))['prelu_slope_vs_own_fp32_plan']
for b,v in d.items(): print(b, 'vec', [round(x,4) for x in v['vec_rel']], 'lstm_bias', [round(x,4) for x in v['lstm_bias_rel_worst']], 'bias max', round(max(v['bias']),3), 'noise', [round(x,3) for x in v['noise']])
"
