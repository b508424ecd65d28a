# small steps: persistent grids of the weight-gradient kernels capped (debug-switch build)
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
export DVAE_HIP_LIB=disentangling-vae_amd/lib/libdvae_hip_debug.so
run() { DVAE_WGRAD_GRID=$1 DVAE_THIN_WGRAD_GRID=$2 python bench.py $3 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3 wgrad_grid=$1 thin_grid=$2', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for g in "256 256" "128 256" "64 256" "128 128" "192 256"; do set -- $g
  for w in "--batch 128" "--batch 256" "--config btcvae_dsprites"; do run $1 $2 "$w"; done
 done
done
