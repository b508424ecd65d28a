# eager vs recorded-plan vs hipGraph issue of the iteration, over batch sizes
python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "replay" 2>&1 | tail -5
for g in eager plan graph; do for b in 64 256 1024; do
  python bench.py --steps 60 --warmup 10 --batch $b --replay $g --no-cpu-baseline --no-roofline --no-parity-check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae 3ch replay=$g B=$b', d['value'], d['ms_per_step'])"
done; done
for g in eager plan graph; do for b in 128 512 2048; do
  python bench.py --steps 60 --warmup 10 --loss factor --batch $b --replay $g --no-cpu-baseline --no-roofline --no-parity-check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor 3ch replay=$g B=$b', d['value'], d['ms_per_step'])"
done; done
