python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "replay or fused_step" 2>&1 | tail -3
for p in 0 1 0 1; do for m in eager plan; do
  DVAE_SIDE_PRIORITY=$p python bench.py --steps 60 --warmup 10 --replay $m --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae B=1024 lowprio=$p replay=$m', d['value'], d['ms_per_step'])"
  DVAE_SIDE_PRIORITY=$p python bench.py --steps 60 --warmup 10 --loss factor --replay $m --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor B=2048 lowprio=$p replay=$m', d['value'], d['ms_per_step'])"
done; done
