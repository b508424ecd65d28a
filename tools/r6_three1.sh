# three-queue weight-gradient schedule: parity of whole steps with it forced on, then A/B
DVAE_DEBUG=1 DVAE_THREE_STREAM_ROWS=100000 timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_sizes.py tests/test_gpu_uint8_input.py -m gpu -q --no-header -x 2>&1 | tail -4
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { DVAE_DEBUG=1 DVAE_THREE_STREAM_ROWS=$1 python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 three_rows=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in 0 100000; do
  for w in "--batch 64" "--batch 128" "--batch 256" "--batch 512" "--batch 1024" "--config btcvae_dsprites" "--config factor_dsprites" "--config factor_celeba"; do run $t "$w"; done
 done
done
