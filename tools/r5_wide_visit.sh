set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_wide_latent.py -q --timeout=120 --no-header -x > gpurun_out/wide_pytest.log 2>&1
echo "wide pytest exit: $?" | tee -a gpurun_out/wide_pytest.log
tail -n 30 gpurun_out/wide_pytest.log | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_fused_core.py -q --timeout=120 --no-header -k "btcvae or reparam or epilogue or other_latent or reference_style or evaluator or fc_chain or autograd" > gpurun_out/regress_pytest.log 2>&1
echo "regress pytest exit: $?" | tee -a gpurun_out/regress_pytest.log
tail -n 8 gpurun_out/regress_pytest.log | cut -c1-300
timeout 200 python tools/wide_latent_time.py 128 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wide_latent_time.txt
