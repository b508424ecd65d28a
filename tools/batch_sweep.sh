for b in 64 128 256 512 1024 2048; do
  python bench.py --steps 30 --warmup 10 --batch $b --no-cpu-baseline --no-roofline --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae 3ch B=$b', d['value'], d['ms_per_step'])"
done
for b in 64 256; do
  python bench.py --steps 30 --warmup 10 --batch $b --channels 1 --no-cpu-baseline --no-roofline --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae 1ch B=$b', d['value'], d['ms_per_step'])"
  python bench.py --steps 30 --warmup 10 --batch $b --channels 1 --loss factor --no-cpu-baseline --no-roofline --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor 1ch B=$b', d['value'], d['ms_per_step'])"
done
