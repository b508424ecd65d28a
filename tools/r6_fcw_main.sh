BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_FCW_MAIN=$1 python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 fcw_main=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in 0 1; do
  for w in "--batch 64" "--batch 128" "--batch 256" "--batch 512" "--batch 1024" "--config btcvae_dsprites" "--config factor_dsprites"; do run $t "$w"; done
 done
done
