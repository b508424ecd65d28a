SA="--steps 60 --warmup 15 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2; do for e in 4718592 100000000; do
 DVAE_DEBUG=1 DVAE_SMALL_SHARD_ELEMS=$e python bench.py --config btcvae_celeba --batch 2048 --shard-world 2 $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('btcvae 1024/rank of 2 small_shard_elems=$e single', d['single_process']['ms_per_step'], 'rccl', d['transports']['rccl']['ms_per_step'])"
 DVAE_DEBUG=1 DVAE_SMALL_SHARD_ELEMS=$e python bench.py --config factor_celeba --shard-world 2 $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('factor tensor 1024/rank of 2 small_shard_elems=$e single', d['single_process']['ms_per_step'], 'rccl', d['transports']['rccl']['ms_per_step'])"
 DVAE_DEBUG=1 DVAE_SMALL_SHARD_ELEMS=$e python bench.py --config factor_celeba --shard-world 4 $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('factor tensor 512/rank of 4 small_shard_elems=$e single', d['single_process']['ms_per_step'], 'rccl', d['transports']['rccl']['ms_per_step'])"
done; done
