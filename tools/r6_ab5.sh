L=disentangling-vae_amd/lib
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mask_bits.py tests/test_gpu_step.py tests/test_gpu_ddp.py -m gpu -q --no-header -x 2>&1 | tail -3
for rep in 1 2; do
  python tools/ab_kernels.py 1024 wg16,wg8,up16
  DVAE_HIP_LIB=$L/libdvae_hip_wgw2.so python tools/ab_kernels.py 1024 wg16,wg8
done
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in "" wgw2; do
  if [ -z "$v" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$v.so; fi
  for b in 1024 128; do python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b lib=${v:-default}', d['ms_per_step'])"; done
 done
 unset DVAE_HIP_LIB
 for b in 1024 128 32; do DVAE_DEBUG=1 DVAE_HEAD_SIDE=0 python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b head on main', d['ms_per_step'])"; done
 python bench.py --batch 32 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=32 default', d['ms_per_step'])"
 for c in btcvae_dsprites factor_dsprites factor_celeba; do
   python bench.py --config $c --steps 80 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check --no-drop-in 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c default', d['ms_per_step'])"
   DVAE_DEBUG=1 DVAE_HEAD_SIDE=0 python bench.py --config $c --steps 80 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check --no-drop-in 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c head on main', d['ms_per_step'])"
 done
done
