# fused 4x4 conv ends inside the FC chain launches: whole GPU suite, then the step with / without (DVAE_DEBUG=1 DVAE_FUSE_ENDS=0)
timeout 1500 python -m pytest tests -m gpu -q --no-header -x --timeout=600 2>&1 | tail -8
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in 1 0; do
  for b in 1024 128; do DVAE_DEBUG=1 DVAE_FUSE_ENDS=$v python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_celeba B=$b fuse_ends=$v', d['ms_per_step'])"; done
  for c in btcvae_dsprites factor_dsprites factor_celeba; do DVAE_DEBUG=1 DVAE_FUSE_ENDS=$v python bench.py --config $c $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c fuse_ends=$v', d['ms_per_step'])"; done
 done
done
