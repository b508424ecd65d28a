timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fused_core.py -m gpu -q --no-header -x 2>&1 | tail -4
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in 1 0; do
  for c in vae_mnist; do DVAE_DEBUG=1 DVAE_FUSE_ENDS=$v python bench.py --config $c $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c fuse_ends=$v', d['ms_per_step'])"; done
 done
done
