BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in "1 0" "0 0" "1 1"; do
  set -- $v
  for b in 1024 256 128; do DVAE_DEBUG=1 DVAE_EARLY_THIN=$1 DVAE_BIG_WGRAD_FIRST=$2 python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae B=$b early_thin=$1 big_first=$2', d['ms_per_step'])"; done
  for c in btcvae_dsprites factor_dsprites factor_celeba; do DVAE_DEBUG=1 DVAE_EARLY_THIN=$1 DVAE_BIG_WGRAD_FIRST=$2 python bench.py --config $c $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c early_thin=$1 big_first=$2', d['ms_per_step'])"; done
 done
done
