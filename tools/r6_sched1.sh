BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in 0 1; do
  DVAE_DEBUG=1 DVAE_EARLY_THIN=$v python bench.py $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_celeba early_thin=$v', d['ms_per_step'])"
  DVAE_DEBUG=1 DVAE_EARLY_THIN=$v python bench.py --config factor_celeba $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factor_celeba early_thin=$v', d['ms_per_step'])"
  DVAE_DEBUG=1 DVAE_EARLY_THIN=$v python bench.py --batch 512 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae B=512 early_thin=$v', d['ms_per_step'])"
 done
done
