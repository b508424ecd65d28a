# FactorVAE: the discriminator's part of the step on the third stream beside the decoder's (DVAE_PAR_DISC_MAX_ROWS): parity forced on, then A/B
DVAE_DEBUG=1 DVAE_PAR_DISC_MAX_ROWS=100000 timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_sizes.py tests/test_gpu_discriminator.py -m gpu -q --no-header -x -k "factor or disc" 2>&1 | tail -3
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_PAR_DISC_MAX_ROWS=$1 python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 par_disc_rows=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in 0 100000; do
  for w in "--config factor_dsprites" "--config factor_dsprites --batch 64" "--config factor_dsprites --batch 512" "--config factor_celeba --batch 256" "--config factor_celeba --batch 512" "--config factor_celeba --batch 1024" "--config factor_celeba"; do run $t "$w"; done
 done
done
