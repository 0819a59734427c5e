cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "DCCRN-3-4000-E" > $O/r5k_test.log 2>&1; tail -3 $O/r5k_test.log | cut -c1-300
for p in 1 2 1 2; do
SEFD_STFT_PPW=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/r5k_bench.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5k_bench.log').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print(d['ms_per_step'], 'stft', k['stft_fft']['ms'], k['stft_fft']['ms_isolated'], k['stft_fft']['frac_algorithmic'], 'istft', k['istft_fft']['ms'])
PY
done
