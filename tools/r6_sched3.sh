BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_EARLY_THIN=$1 DVAE_TAIL_MAIN=$2 python bench.py $3 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3 early_thin=$1 tail_main=$2', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for e in 1 2; do for t in default conv3 conv_64 none; do
  for w in "--batch 128" "--batch 256" "--config btcvae_dsprites" "--config factor_dsprites"; do run $e $t "$w"; done
 done; done
 for e in 1 2; do run $e default "--batch 1024"; run $e default "--config factor_celeba"; done
done
