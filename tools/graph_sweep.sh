python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "hip_graph or evaluator or factor_step" 2>&1 | tail -15
python tools/host_profile.py btcvae 64 2>&1 | head -60
for g in 0 1; do for b in 64 256 1024; do
  python bench.py --steps 40 --warmup 10 --batch $b --graph $g --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae 3ch graph=$g B=$b', d['value'], d['ms_per_step'])"
done; done
for g in 0 1; do for b in 128 2048; do
  python bench.py --steps 40 --warmup 10 --loss factor --batch $b --graph $g --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor 3ch graph=$g B=$b', d['value'], d['ms_per_step'])"
done; done
