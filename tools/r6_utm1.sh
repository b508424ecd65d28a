# round 6, second session: where k_up_thin_mm's time goes (ablation builds) and load-placement / barrier variants
L=disentangling-vae_amd/lib
for v in v4 v5 v6 v7; do
  echo "== parity, variant $v"; DVAE_HIP_LIB=$L/libdvae_hip_utm_$v.so timeout 600 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x -k "convT3 or convT_sigmoid or saturated" 2>&1 | tail -2
done
for rep in 1 2; do
  python tools/ab_kernels.py 1024 utm
  for v in v4 v5 v6 v7 a1 a2 a4 a8 a16 a6 a30; do DVAE_HIP_LIB=$L/libdvae_hip_utm_$v.so python tools/ab_kernels.py 1024 utm; done
done
python tools/ab_kernels.py 128 utm
for v in v4 v5 v7; do DVAE_HIP_LIB=$L/libdvae_hip_utm_$v.so python tools/ab_kernels.py 128 utm; done
