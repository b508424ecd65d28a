BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_EARLY_THIN=$1 DVAE_TAIL_MAIN=$2 python bench.py $3 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3 early_thin=$1 tail_main=$2', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for e in 1 2; do for w in "--batch 32" "--batch 64" "--batch 96" "--batch 128"; do run $e default "$w"; done; done
 for t in default conv2,conv3,conv_64 conv3 conv_64; do for w in "--batch 256" "--config btcvae_dsprites" "--batch 192"; do run 1 $t "$w"; done; done
done
