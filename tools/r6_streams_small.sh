BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_SINGLE_STREAM_ELEMS=$1 python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 single_stream_elems=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in 196608 0; do
  for w in "--config vae_mnist" "--config vae_mnist --batch 128" "--config vae_mnist --batch 16" "--batch 4" "--batch 8" "--batch 16"; do run $t "$w"; done
 done
done
