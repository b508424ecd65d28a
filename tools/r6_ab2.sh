L=disentangling-vae_amd/lib
for rep in 1 2; do
  python tools/ab_kernels.py 1024 wg16,wg8,utm
  DVAE_HIP_LIB=$L/libdvae_hip_wgold.so python tools/ab_kernels.py 1024 wg16,wg8
  DVAE_HIP_LIB=$L/libdvae_hip_utm1.so python tools/ab_kernels.py 1024 utm
  DVAE_HIP_LIB=$L/libdvae_hip_utmlate.so python tools/ab_kernels.py 1024 utm
done
python tools/ab_kernels.py 128 wg16,wg8,utm
DVAE_HIP_LIB=$L/libdvae_hip_wgold.so python tools/ab_kernels.py 128 wg16,wg8
DVAE_HIP_LIB=$L/libdvae_hip_utm1.so python tools/ab_kernels.py 128 utm
timeout 600 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x 2>&1 | tail -3
DVAE_HIP_LIB=$L/libdvae_hip_utm1.so timeout 600 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x 2>&1 | tail -3
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2; do
 for v in "" wgold wgb128 utm1; do
  if [ -z "$v" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$v.so; fi
  for b in 1024 128; do python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b lib=${v:-default}', d['ms_per_step'])"; done
 done
done
