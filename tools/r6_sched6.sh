SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2 3; do for e in auto 1 2; do DVAE_DEBUG=1 DVAE_EARLY_THIN=$e python bench.py --config btcvae_celeba $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('shard early_thin=$e single', d['single_process']['ms_per_step']); print('shard early_thin=$e rccl', d['transports']['rccl']['ms_per_step'])"; done; done
