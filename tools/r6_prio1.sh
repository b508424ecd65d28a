BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
L=disentangling-vae_amd/lib
run() { if [ "$1" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$1.so; fi; python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 lib=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in default lowprio; do
  for w in "--batch 64" "--batch 128" "--batch 256" "--batch 1024" "--config btcvae_dsprites" "--config factor_dsprites" "--config factor_celeba"; do run $t "$w"; done
 done
done
