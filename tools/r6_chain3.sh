# fused conv ends, direct LDS hand-over: parity, kernel timing, the step around the row limit
timeout 900 python -m pytest tests/test_gpu_fused_core.py tests/test_gpu_step.py -m gpu -q --no-header -x 2>&1 | tail -4
python tools/ab_chain_ends.py 128 256 512
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in 1 0; do
  for b in 512 256 128 64; do DVAE_DEBUG=1 DVAE_FUSE_ENDS=$v python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_celeba B=$b fuse_ends=$v', d['ms_per_step'])"; done
  for c in btcvae_dsprites factor_dsprites; do DVAE_DEBUG=1 DVAE_FUSE_ENDS=$v python bench.py --config $c $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c fuse_ends=$v', d['ms_per_step'])"; done
 done
done
