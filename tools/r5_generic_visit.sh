# the split generic weight gradient (32x32 MNIST geometry): kernel tests, whole-step tests of the 32x32 configurations, bench leg
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_bench_sizes.py tests/test_gpu_uint8_input.py -q --timeout=120 --no-header -k "conv or fused_step_vs_oracle or mnist or 32" > gpurun_out/generic_pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/generic_pytest.log
tail -n 12 gpurun_out/generic_pytest.log | cut -c1-300
for r in 1 2; do
timeout 120 python bench.py --config vae_mnist --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-extra-configs --no-drop-in 2>&1 | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vae_mnist', d['value'], d['ms_per_step'], d.get('parity_check',{}).get('ok'))" | tee -a gpurun_out/generic_bench.txt
done
