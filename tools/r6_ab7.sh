L=disentangling-vae_amd/lib
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fused_core.py tests/test_gpu_bench_sizes.py tests/test_gpu_mask_bits.py -m gpu -q --no-header -x 2>&1 | tail -3
for rep in 1 2; do
  python tools/ab_kernels.py 1024 down16,down16m,up16,up16m,wg16,wg8
  DVAE_HIP_LIB=$L/libdvae_hip_bold.so python tools/ab_kernels.py 1024 down16,down16m,up16,up16m,wg16,wg8
done
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in "" bold; do
  if [ -z "$v" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$v.so; fi
  for b in 1024 128; do python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b lib=${v:-default}', d['ms_per_step'])"; done
 done
done
