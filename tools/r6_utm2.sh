# round 6, second session: target-load placement variants of k_up_thin_mm alone and inside the step; the step without the estimator
L=disentangling-vae_amd/lib
for v in v13 v21; do
  echo "== parity, variant $v"; DVAE_HIP_LIB=$L/libdvae_hip_utm_$v.so timeout 600 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x -k "convT3 or convT_sigmoid or saturated" 2>&1 | tail -2
done
for rep in 1 2; do
  python tools/ab_kernels.py 1024 utm
  for v in v5 v13 v21; do DVAE_HIP_LIB=$L/libdvae_hip_utm_$v.so python tools/ab_kernels.py 1024 utm; done
done
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in "" utm_v5 utm_v13 utm_v21; do
  if [ -z "$v" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$v.so; fi
  for b in 1024 128; do python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b lib=${v:-default}', d['ms_per_step'])"; done
 done
done
unset DVAE_HIP_LIB
for l in btcvae betaH; do for b in 1024 256 128; do python bench.py --loss $l --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b loss=$l', d['ms_per_step'])"; done; done
