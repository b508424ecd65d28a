python -m pytest tests/test_gpu_step.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -3
for cap in 256 128 192 256 128; do
  DVAE_WGRAD_CAP=$cap python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('btcvae cap=$cap', d['value'], d['ms_per_step'])"
done
for cap in 256 128; do
  DVAE_WGRAD_CAP=$cap python bench.py --steps 60 --warmup 10 --loss factor --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('factor cap=$cap', d['value'], d['ms_per_step'])"
done
bash tools/ab.sh DVAE_GEMM_FC
