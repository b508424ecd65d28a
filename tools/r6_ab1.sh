L=disentangling-vae_amd/lib
for rep in 1 2; do
  python tools/ab_kernels.py 1024 wg16,wg8,utm
  DVAE_HIP_LIB=$L/libdvae_hip_wgb128.so python tools/ab_kernels.py 1024 wg16,wg8
  DVAE_HIP_LIB=$L/libdvae_hip_utmlate.so python tools/ab_kernels.py 1024 utm
done
python tools/ab_kernels.py 128 wg16,wg8,utm
DVAE_HIP_LIB=$L/libdvae_hip_wgb128.so python tools/ab_kernels.py 128 wg16,wg8
DVAE_HIP_LIB=$L/libdvae_hip_utmlate.so python tools/ab_kernels.py 128 utm
timeout 600 python -m pytest tests/test_gpu_fused_core.py tests/test_gpu_kernels.py -m gpu -q --no-header -x 2>&1 | tail -3
DVAE_HIP_LIB=$L/libdvae_hip_wgb128.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py -m gpu -q --no-header -x -k "wgrad or step" 2>&1 | tail -3
