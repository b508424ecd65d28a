SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which rccl"
for rep in 1 2 3; do for e in 4718592 100000000; do for w in 2 1; do DVAE_DEBUG=1 DVAE_SMALL_SHARD_ELEMS=$e python bench.py --config btcvae_celeba --shard-world $w $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('world=$w small_shard_elems=$e rccl', d['transports']['rccl']['ms_per_step'])"; done; done; done
