# where does a capped weight-gradient grid pay?  (debug-switch build)
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
export DVAE_HIP_LIB=disentangling-vae_amd/lib/libdvae_hip_debug.so
run() { DVAE_WGRAD_GRID=$1 python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 wgrad_grid=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for g in 256 224 192 160 128; do
  for w in "--batch 192" "--batch 256" "--batch 384" "--batch 512" "--config factor_dsprites"; do run $g "$w"; done
 done
 for g in 256 224 192; do run $g "--batch 1024"; done
done
