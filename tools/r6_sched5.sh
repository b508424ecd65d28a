DVAE_DEBUG=1 DVAE_EARLY_THIN=2 timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_sizes.py -m gpu -q --no-header -x 2>&1 | tail -3
BA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2 3; do for e in 1 0; do DVAE_DEBUG=1 DVAE_EARLY_THIN=$e python bench.py --config btcvae_celeba $BA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('shard early_thin=$e(1 = mode 2 at 128 rows) single', d['single_process']['ms_per_step']); print('shard early_thin=$e(1 = mode 2 at 128 rows) rccl', d['transports']['rccl']['ms_per_step'])"; done; done
